"""CPU: the C-ABI shared library loads and exports every symbol include/v3d.h declares (no compute
calls without a GPU), and the product path fails loudly without a device."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT, v3d


def _declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'v3d.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(v3d_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported_and_bound():
    lib_mod = v3d('_lib')
    if not os.path.exists(lib_mod.LIB_PATH):
        v3d('build').build()
    names = _declared_symbols()
    assert len(names) >= 9
    raw = ctypes.CDLL(lib_mod.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), 'library does not export %s' % n
        assert n in lib_mod.SIGNATURES, 'no ctypes signature for %s' % n
    assert set(lib_mod.SIGNATURES) == set(names)
    lib = lib_mod.load()
    assert lib.v3d_version() == lib_mod.ABI_VERSION


def test_host_side_argument_validation():
    """Error paths that return before touching the device."""
    lib_mod = v3d('_lib')
    lib = lib_mod.load()
    # zero-bordered channel-last copy of the features ((Hf + 2) x (Wf + 2) cells per image + a zero row of Wf + 3 cells behind
    # the last one) + one 36-float camera block per image, each 256-byte aligned
    # bordered maps + the zero tail (two bordered rows + 16 cells: the window kernel copies whole 8-cell runs) + camera blocks
    assert lib.v3d_psv_workspace_bytes(8, 32, 64, 80) == (8 * 66 * 82 + 2 * 82 + 16) * 32 * 4 + 1280
    rc = lib.v3d_psv_variance_f32(None, None, None, None, None, None, None, 1, 1, 1, 32, 4, 4, 8, 8,
                                  0.5, 0.05, 8, 8, 8, None, None, 0, None)
    assert rc == -2 and b'null' in lib.v3d_last_error()
    with pytest.raises(lib_mod.V3DLibraryError):
        lib_mod.check(rc, 'psv')


def test_no_cpu_fallback():
    mvs = v3d('mvsnet')
    lib_mod = v3d('_lib')
    net = mvs.CostRegNet(32, 8).eval()
    with pytest.raises(lib_mod.V3DLibraryError):
        net(torch.zeros(1, 32, 8, 8, 8))
    with pytest.raises(lib_mod.V3DLibraryError):
        mvs.plane_sweep_variance(torch.zeros(2, 32, 4, 4), torch.eye(3).repeat(2, 1, 1),
                                 torch.zeros(2, 3), torch.eye(3).repeat(2, 1, 1),
                                 torch.tensor([[0, 0], [0, 1]]), 0.5, 0.05, 8, (16, 16), (8, 8))


def test_state_dict_keys_match_reference_naming():
    """CostRegNet keys enumerated in SURVEY.md §8b (mvsnet.py:133-163)."""
    mvs, syn = v3d('mvsnet'), v3d('synthetic')
    net = mvs.CostRegNet(32, 8)
    sd = syn.costregnet_weights()
    res = net.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    assert all(k.endswith('num_batches_tracked') for k in res.missing_keys)
    assert net.conv7.deconv.weight.shape == (64, 32, 3, 3, 3)
    assert net.prob.weight.shape == (1, 8, 3, 3, 3)


def test_edges_to_csr_matches_unique_semantics():
    mvs = v3d('mvsnet')
    e = torch.tensor([[5, 2, 5, 2, 9, 5], [0, 1, 2, 3, 4, 5]])
    ref_idx, ref_img, ofs, src = mvs.edges_to_csr(e)
    assert ref_idx.tolist() == [2, 5, 9]
    assert ofs.tolist() == [0, 2, 5, 6]
    assert src.tolist() == [1, 3, 0, 2, 5, 4]


def test_pl3dvnet_default_construction_and_unsupported_feat_dim():
    """The reference's default construction PL3DVNet(depth_train, depth_test, edge_len) builds what the reference's own
    signature builds (lightningmodel.py:18: feat_dim=16, module shapes as :34-43 gives them); feat_dim=32 -- the value of
    mv3d/config.py:42 and of the released checkpoints -- builds the network the fast kernels are specialised for; any other
    width is rejected at construction time instead of failing on an assert deep inside SparseUNet."""
    lm = v3d('lightningmodel')
    net = lm.PL3DVNet(None, {'size': (8, 8)}, 0.08, feat_dim=32)
    assert net.hparams.feat_dim == 32 and net.sparse_conv.dims == (64, 128, 128)
    n16 = lm.PL3DVNet(None, {'size': (8, 8)}, 0.08)
    assert n16.hparams.feat_dim == 16
    assert n16.sparse_conv.dims == (32, 128, 128) and n16.mvsnet.cnn_3d.conv0.conv.in_channels == 16
    assert n16.decoder.in_dim == 128 + 128 + 48 and n16.refine_half.in_dim == 17
    with pytest.raises(ValueError, match='feat_dim'):
        lm.PL3DVNet(None, {'size': (8, 8)}, 0.08, feat_dim=24)


def test_isa_guard_of_the_fused_decoder():
    """3dvnet_amd/isa_check.py (run by 3dvnet_amd/build.py on the default build): decoder_fused_kernel in the built object has no
    scratch instruction, its LDS read signature equals the pinned one, and a moved signature fails the check with the pinned
    compiler (and only warns with another one)."""
    import importlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tag = open(os.path.join(root, '3dvnet_amd', 'build', 'linked_flags')).read().strip()
    obj = os.path.join(root, '3dvnet_amd', 'build', tag, 'decoder.o')
    if not os.path.exists(obj) or not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-objdump'):
        pytest.skip('no decoder.o / llvm tools here')
    chk = importlib.import_module('3dvnet_amd.isa_check')
    sig, scratch = chk.signature(obj)
    assert scratch == 0
    assert sig['ds_read_b128'] > 0
    assert chk.check(obj, strict=True) == sig
    pinned, comp = chk.PINNED, chk.PINNED_COMPILER
    try:
        if pinned is not None:
            chk.PINNED = dict(pinned, ds_read_b64=pinned.get('ds_read_b64', 0) + 1)
            chk.PINNED_COMPILER = chk.compiler_id()
            with pytest.raises(RuntimeError, match='signature'):
                chk.check(obj)
            chk.PINNED_COMPILER = 'some other compiler'
            assert chk.check(obj) == sig                    # warns, does not fail
        with pytest.raises(RuntimeError, match='could not run'):
            chk.check(obj + '.missing', strict=True)
        assert chk.check(obj + '.missing', strict=False) is None
    finally:
        chk.PINNED, chk.PINNED_COMPILER = pinned, comp

"""CPU checks of arithmetic identities the HIP kernels rely on (numpy emulation of the fp32 instruction sequences)."""
import importlib
import os

import numpy as np
import pytest
import torch


def _fma32(a, b, c):
    """fmaf for float32 arrays: the product of two float32 is exact in float64, one rounding to float32 at the end
    (the float64 addition is exact or far below a float32 half-ulp for the magnitudes used here)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def test_uniform_division_by_reciprocal_is_exact():
    """psv_variance_reuse_kernel divides by the wave-uniform W - 1 / H - 1 as q0 = x * rc, q1 = fma(fma(-q0, c, x), rc, q0)
    with rc = RN(1 / c).  Markstein: q1 is the correctly rounded x / c unless c's significand is all ones.  Checked against
    IEEE float32 division for image extents from 4 to 4096 and operands from 1e-3 to 1e7 (coordinates of points far
    behind a camera included)."""
    rng = np.random.default_rng(0)
    for c in (3, 7, 63, 79, 255, 319, 479, 639, 959, 1079, 1279, 1919, 4095):
        c32 = np.float32(c)
        rc = np.float32(1.0 / np.float64(c))
        x = (rng.standard_normal(400_000) * rng.choice([1e-3, 1.0, 50.0, 400.0, 1e4, 1e7], 400_000)).astype(np.float32)
        q0 = (x * rc).astype(np.float32)
        q1 = _fma32(_fma32(-q0, np.full_like(x, c32), x), np.full_like(x, rc), q0)
        assert np.array_equal(q1, (x / c32).astype(np.float32)), c


def test_view_count_mean_by_reciprocal_is_exact():
    """The window warp kernel's mean over a view count that is not a power of two (cfg5: 11 edges) is the same Markstein
    sequence with c = the count: q0 = x * rc, q = fma(fma(-q0, c, x), rc, q0), rc = RN(1 / c) -- equal to the IEEE quotient
    x / c (what torch_scatter's mean computes) for every normal-range sum; checked for counts 3 .. 31 on sums and sums of
    squares spanning 1e-12 .. 1e8 (zeros included: 0 / c = 0 either way)."""
    rng = np.random.default_rng(2)
    for c in [3, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15, 17, 19, 21, 23, 27, 31]:
        c32 = np.float32(c)
        rc = np.float32(1.0) / c32
        x = (rng.standard_normal(300_000) * rng.choice([1e-12, 1e-6, 1e-3, 1.0, 11.0, 300.0, 1e4, 1e8], 300_000)).astype(np.float32)
        x[::1000] = 0.0
        q0 = (x * rc).astype(np.float32)
        q1 = _fma32(_fma32(-q0, np.full_like(x, c32), x), np.full_like(x, rc), q0)
        assert np.array_equal(q1, (x / c32).astype(np.float32)), c


def test_power_of_two_mean_is_a_multiplication():
    """torch_scatter mean = sum / count; for a power-of-two count the kernels multiply by 1 / count instead."""
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(200_000) * rng.choice([1e-6, 1e-2, 1.0, 30.0], 200_000)).astype(np.float32)
    for n in (1, 2, 4, 8, 16):
        assert np.array_equal((x / np.float32(n)).astype(np.float32), (x * np.float32(1.0 / n)).astype(np.float32))


def test_bf16_split_carries_sixteen_mantissa_bits():
    """x = hi + lo with hi = RNE_bf16(x), lo = RNE_bf16(x - hi): |x - (hi + lo)| <= 2^-17 |x| (the operand precision of the
    split-bf16 MFMA kernels), and hi + lo is exactly representable in float32."""
    def rne_bf16(v):
        u = v.view(np.uint32).astype(np.uint64)
        return ((((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16) & 0xffffffff).astype(np.uint32).view(np.float32)
    rng = np.random.default_rng(2)
    x = (rng.standard_normal(300_000) * rng.choice([1e-4, 1.0, 100.0], 300_000)).astype(np.float32)
    hi = rne_bf16(x)
    lo = rne_bf16((x - hi).astype(np.float32))
    rec = (hi.astype(np.float64) + lo.astype(np.float64))
    assert np.array_equal(rec.astype(np.float32).astype(np.float64), rec)
    assert np.all(np.abs(rec - x.astype(np.float64)) <= 2.0 ** -17 * np.abs(x.astype(np.float64)) + 1e-45)


def _deconv_cells_1d(x, w, shifted):
    """ConvTranspose1d(k=3, stride 2, pad 1, output_padding 1) written as the cell decomposition the HIP kernels use.
    Aligned cells (deconvg_bf16x2_kernel): cell j = outputs (2j, 2j+1) from inputs (j, j+1):
        out[2j] = in[j] w[1];  out[2j+1] = in[j] w[2] + in[j+1] w[0]
    Shifted cells (conv9_prob_kernel, whose halo'd tile starts at an odd output): cell j = outputs (2j+1, 2j+2):
        out[2j+1] = in[j] w[2] + in[j+1] w[0];  out[2j+2] = in[j+1] w[1]"""
    n = x.shape[0]
    out = np.zeros(2 * n, dtype=np.float64)
    xin = lambda i: x[i] if 0 <= i < n else 0.0
    if not shifted:
        for j in range(n):
            out[2 * j] = xin(j) * w[1]
            out[2 * j + 1] = xin(j) * w[2] + xin(j + 1) * w[0]
    else:
        for j in range(-1, n):
            if 0 <= 2 * j + 1 < 2 * n:
                out[2 * j + 1] = xin(j) * w[2] + xin(j + 1) * w[0]
            if 0 <= 2 * j + 2 < 2 * n:
                out[2 * j + 2] = xin(j + 1) * w[1]
    return out


def test_transposed_conv_cell_decomposition_matches_torch():
    """The tap tables packed by v3d_costreg_pack for conv7 / conv8 / conv9 (kernel tap of (output parity, input offset))
    restate torch's ConvTranspose(k 3, s 2, p 1, output_padding 1); the 3D layers are the tensor product of this 1D map."""
    import torch
    rng = np.random.default_rng(3)
    for n in (1, 2, 5, 8):
        x = rng.standard_normal(n)
        w = rng.standard_normal(3)
        ref = torch.nn.functional.conv_transpose1d(torch.tensor(x).view(1, 1, n), torch.tensor(w).view(1, 1, 3), stride=2,
                                                   padding=1, output_padding=1).view(-1).numpy()
        for shifted in (False, True):
            np.testing.assert_allclose(_deconv_cells_1d(x, w, shifted), ref, rtol=0, atol=1e-12)


def test_stride_two_conv_as_four_tap_k_steps():
    """convg_bf16x2_kernel evaluates a k=3 conv with K steps of 4 x taps x 8 channels, the 4th tap carrying zero weights,
    reading input x = S * out_x + tap - 1 (S = 1 or 2): check the index map against torch conv1d."""
    import torch
    rng = np.random.default_rng(4)
    for stride in (1, 2):
        n = 11
        x = rng.standard_normal(n)
        w = rng.standard_normal(3)
        ref = torch.nn.functional.conv1d(torch.tensor(x).view(1, 1, n), torch.tensor(w).view(1, 1, 3), stride=stride,
                                         padding=1).view(-1).numpy()
        w4 = np.concatenate([w, [0.0]])
        xin = lambda i: x[i] if 0 <= i < n else 0.0
        out = np.array([sum(w4[t] * xin(stride * o + t - 1) for t in range(4)) for o in range(len(ref))])
        np.testing.assert_allclose(out, ref, rtol=0, atol=1e-12)


def test_product_slice_edges_matches_reference_golden():
    """Row H1 on the PRODUCT's copy (3dvnet_amd/utils.py::slice_edges, what process_scene calls): the reference's
    utils.slice_edges output captured in the H_misc golden, plus order preservation on a shuffled edge list."""
    import numpy as np
    import torch
    from conftest import v3d
    from helpers import load_golden, t
    utils = v3d('utils')
    g = load_golden('H_misc')
    assert np.array_equal(utils.slice_edges(t(g['edges']), 3, 5, 0).numpy(), g['sliced'])
    e = torch.tensor([[5, 2, 9, 2, 7, 5], [1, 2, 3, 4, 5, 6]])
    assert utils.slice_edges(e, 2, 6, 0).tolist() == [[5, 2, 2, 5], [1, 2, 4, 6]]
    assert utils.slice_edges(e, 3, 6, 1).tolist() == [[9, 2, 7], [3, 4, 5]]
    assert utils.slice_edges(e, 10, 12, 0).shape == (2, 0)


def test_pack_cache_key_sees_replaced_and_moved_parameters():
    """the packed-weight caches must not be keyed on tensor._version alone -- p.data = ..., assign-style
    loads and device moves create new storage with coinciding versions."""
    import torch
    from conftest import v3d
    mvs = v3d('mvsnet')
    m = torch.nn.Linear(4, 4)
    k0 = mvs.module_state_key(m)
    assert k0 == mvs.module_state_key(m)
    with torch.no_grad():
        m.weight.data = m.weight.data.clone()               # same version counter, new storage
    k1 = mvs.module_state_key(m)
    assert k1 != k0
    m.weight = torch.nn.Parameter(m.weight.detach().clone())    # new Parameter object
    k2 = mvs.module_state_key(m)
    assert k2 != k1
    with torch.no_grad():
        m.bias.add_(1.0)                                    # in-place update bumps the version
    assert mvs.module_state_key(m) != k2
    m.load_state_dict({k: v.clone() for k, v in m.state_dict().items()}, assign=True)
    assert mvs.module_state_key(m) != k2


def test_whole_model_state_dict_keys_and_checkpoint_round_trip(tmp_path):
    """``PL3DVNet.state_dict()`` against the reference's key list (tests/golden/H_state_dict_keys.json, generated by importing the
    reference's modules: CostRegNet, PointNet, the decoder's Conv1d stack, the three PropagationNets -- names AND shapes), the
    MinkowskiEngine naming of SURVEY.md 8b for ``sparse_conv.*`` (``.kernel`` [27, Cin, Cout] / [Cin, Cout], ``.gn.weight / .gn.bias``,
    no biases), and a Lightning-style checkpoint ({'state_dict', 'hyper_parameters'}) through ``PL3DVNet.load_from_checkpoint``
    (mv3d/eval-3dvnet.py:134): every tensor comes back bit for bit."""
    import json
    import os
    lm = importlib.import_module('3dvnet_amd.lightningmodel')
    hp = dict(depth_train={'size': (56, 56)}, depth_test={'size': (56, 56)}, edge_len=0.08, feat_dim=32, img_size=(256, 320))
    net = lm.PL3DVNet(**hp)
    sd = net.state_dict()
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'H_state_dict_keys.json')))
    for k, shape in gold.items():
        assert k in sd, 'reference key %s missing' % k
        assert list(sd[k].shape) == shape, (k, list(sd[k].shape), shape)
    ours = {k for k in sd if k.split('.')[0] in ('pointnet', 'decoder', 'refine_quarter', 'refine_half', 'refine_full')
            or k.startswith('mvsnet.cnn_3d.')}
    assert ours == set(gold), sorted(ours ^ set(gold))[:10]
    sparse = {k: v for k, v in sd.items() if k.startswith('sparse_conv.')}
    assert sparse and all(k.endswith(('.kernel', '.gn.weight', '.gn.bias')) for k in sparse), \
        [k for k in sparse if not k.endswith(('.kernel', '.gn.weight', '.gn.bias'))][:5]
    for k, v in sparse.items():
        if k.endswith('.kernel'):
            assert (v.dim() == 3 and v.shape[0] == 27) or v.dim() == 2, (k, tuple(v.shape))
    assert set(k.split('.')[0] for k in sd) == {'mvsnet', 'pointnet', 'sparse_conv', 'decoder', 'refine_quarter', 'refine_half',
                                                'refine_full'}
    g = torch.Generator().manual_seed(5)
    rnd = {k: (torch.rand(v.shape, generator=g) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    path = str(tmp_path / 'model.ckpt')
    torch.save({'state_dict': rnd, 'hyper_parameters': hp, 'epoch': 3}, path)
    net2 = lm.PL3DVNet.load_from_checkpoint(path)
    assert not net2.training and net2.hparams.edge_len == 0.08
    for k, v in net2.state_dict().items():
        assert torch.equal(v, rnd[k]), k
    bad = dict(rnd)
    bad['decoder.net.9.weight'] = torch.zeros(1)
    torch.save({'state_dict': bad, 'hyper_parameters': hp}, path)
    with pytest.raises(RuntimeError, match='unexpected'):
        lm.PL3DVNet.load_from_checkpoint(path)


def test_traffic_json_aggregates_kernel_families(tmp_path):
    """profiles/make_traffic.py: single kernels by substring, kernel families (the sparse-convolution pipeline's instances, the FLAT
    conv launches of stage 3) as the dispatch-weighted mean; the gfx950 x2 on FETCH_SIZE applied to every entry."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    head = 'kernel,counter,dispatches,avg_value_per_dispatch\n'
    fetch = head + 'decoder_fused_kernel,FETCH_SIZE,6,100.0\n"gemm_gather_pipe_kernel<2, 4, 8, 2, 1>",FETCH_SIZE,20,10.0\n' \
                   '"gemm_gather_pipe_kernel<2, 2, 4, 2, 2>",FETCH_SIZE,10,40.0\n' \
                   '"convg_bf16x2_kernel<CG<32, 32, 1, 14, 2, true> >",FETCH_SIZE,6,300.0\n' \
                   '"convg_bf16x2_kernel<CG<32, 32, 1, 14, 2, false> >",FETCH_SIZE,1,7.0\n'
    write = fetch.replace('FETCH_SIZE', 'WRITE_SIZE')
    (tmp_path / 'f.csv').write_text(fetch)
    (tmp_path / 'w.csv').write_text(write)
    out = tmp_path / 't.json'
    subprocess.check_call([sys.executable, os.path.join(root, 'profiles', 'make_traffic.py'), str(tmp_path / 'f.csv'),
                           str(tmp_path / 'w.csv'), '64', str(out)])
    k = json.load(open(out))['kernels']
    assert k['decoder_fused']['hbm_bytes'] == (2 * 100.0 + 100.0) * 1024
    assert k['sparse_conv_gemm']['fetch_kb_raw'] == pytest.approx((20 * 10.0 + 10 * 40.0) / 30)
    assert k['propagation_conv']['fetch_kb_raw'] == 300.0 and k['costreg_conv4']['fetch_kb_raw'] == 7.0



def test_level_info_behaves_like_the_plain_seven_key_dict():
    """SparseUNet.forward returns one dict per level with keys feats, pts, res, batch, idx, stride, sparse (scenemodeling.py:
    210-237); the package derives pts / idx / batch on first access, and every whole-dict view must see them."""
    import types
    sm = importlib.import_module('3dvnet_amd.scenemodeling')
    lvl = types.SimpleNamespace(coords=torch.tensor([[0, 1, 2, 3], [0, 4, 5, 6]]))
    x = sm.LevelInfo(lvl, 0.5, torch.tensor([[1., 2., 3.]]), torch.zeros(1, dtype=torch.long))
    x['feats'], x['res'], x['stride'], x['sparse'] = torch.zeros(2, 4), 0.5, 1, lvl
    want = ['batch', 'feats', 'idx', 'pts', 'res', 'sparse', 'stride']
    assert len(x) == 7 and 'pts' in x and x.get('nope', 7) == 7
    assert sorted(x.keys()) == want and sorted(dict(x)) == want and sorted({**x}) == want and sorted(x.copy()) == want
    assert sorted(k for k in x) == want and sorted(k for k, _ in x.items()) == want and len(list(x.values())) == 7
    assert torch.equal(x.get('pts'), torch.tensor([[1.5, 3., 4.5], [3., 4.5, 6.]])) and x['batch'].dtype == torch.long
    moved = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in x.items()}
    assert torch.equal(moved['idx'], torch.tensor([[1, 2, 3], [4, 5, 6]]))

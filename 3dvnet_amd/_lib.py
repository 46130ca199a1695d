"""ctypes binding of lib3dvnet_hip.so (C ABI declared in include/v3d.h).

The product path has no CPU fallback: if the shared library is missing or fails to load, every
operator raises ``V3DLibraryError``.  ``load()`` never builds implicitly -- run
``python 3dvnet_amd/build.py`` (or ``__graft_entry__.build()``) first.
"""
import ctypes
import os

import torch  # noqa: F401  -- must be imported BEFORE the dlopen below: the library binds to the
#               HIP runtime (libamdhip64) that PyTorch already loaded; loading ours first would
#               bring a second, uninitialised runtime into the process.

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib3dvnet_hip.so')
ABI_VERSION = 6
PRECISION = {'split_bf16': 0, 'fp32': 1}      # V3D_PRECISION_* of include/v3d.h


def precision_code(name):
    try:
        return PRECISION[name]
    except KeyError:
        raise ValueError("precision must be 'split_bf16' or 'fp32', got %r" % (name,))


c_void_p, c_int, c_float, c_double, c_size_t = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                                ctypes.c_double, ctypes.c_size_t)
c_float_p = ctypes.POINTER(ctypes.c_float)
c_float_pp = ctypes.POINTER(c_float_p)

# name -> (restype, argtypes); mirrors include/v3d.h exactly (tests/test_cabi.py checks the list)
SIGNATURES = {
    'v3d_version': (c_int, []),
    'v3d_last_error': (ctypes.c_char_p, []),
    'v3d_set_option': (c_int, [ctypes.c_char_p, c_int]),
    'v3d_get_option': (c_int, [ctypes.c_char_p, ctypes.POINTER(c_int)]),
    'v3d_timing_enable': (c_int, [c_int]),
    'v3d_timing_collect': (c_int, [c_int, ctypes.c_char_p, c_int, ctypes.POINTER(c_float),
                                   ctypes.POINTER(c_int)]),
    'v3d_psv_workspace_bytes': (c_size_t, [c_int] * 4),
    'v3d_psv_variance_f32': (c_int, [c_void_p] * 7 + [c_int] * 8 + [c_double] * 2 + [c_int] * 3 +
                             [c_void_p, c_void_p, c_size_t, c_void_p]),
    'v3d_psv_sample_positions_f32': (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_double] * 2 + [c_int] * 3 +
                                     [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'v3d_psv_variance_split': (c_int, [c_void_p] * 7 + [c_int] * 8 + [c_double] * 2 + [c_int] * 3 +
                               [c_void_p, c_void_p, c_size_t, c_void_p]),
    'v3d_psv_variance_cl8': (c_int, [c_void_p] * 7 + [c_int] * 8 + [c_double] * 2 + [c_int] * 3 +
                             [c_void_p, c_void_p, c_size_t, c_void_p]),
    'v3d_costreg_depth_cl8': (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p, c_void_p, c_void_p,
                                                                    c_size_t, c_void_p]),
    'v3d_costreg_pack': (c_int, [c_float_pp] * 5 + [c_float_p, c_float_p, c_int, c_int, c_float,
                                                     ctypes.POINTER(c_void_p)]),
    'v3d_costreg_free': (None, [c_void_p]),
    'v3d_costreg_workspace_bytes': (c_size_t, [c_void_p] + [c_int] * 4),
    'v3d_costreg_depth_f32': (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p, c_void_p, c_int, c_void_p,
                                                                    c_size_t, c_void_p]),
    'v3d_costreg_depth_split': (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p, c_void_p, c_void_p,
                                                                      c_size_t, c_void_p]),
    'v3d_costreg_layer_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 4 +
                              [c_void_p, c_int, c_void_p]),
    'v3d_costreg_layer_split_workspace_bytes': (c_size_t, [c_int] * 5),
    'v3d_costreg_layer_split_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 4 +
                                    [c_void_p, c_void_p, c_size_t, c_void_p]),
    'v3d_propagation_pack': (c_int, [c_float_pp] * 5 + [c_int, c_int, c_float, ctypes.POINTER(c_void_p)]),
    'v3d_propagation_free': (None, [c_void_p]),
    'v3d_propagation_workspace_bytes': (c_size_t, [c_void_p, c_int, c_int, c_int]),
    'v3d_propagation_f32': (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 4 + [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'v3d_propagation_up_f32': (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'v3d_backproject_workspace_bytes': (c_size_t, [c_int] * 4),
    'v3d_backproject_variance_f32': (c_int, [c_void_p] * 8 + [c_int] * 10 + [c_double, c_int] +
                                     [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'v3d_gemm_pack': (c_int, [c_float_p, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong,
                              c_int, c_int, c_int, c_float_p, c_float_p, c_float_p, c_float_p,
                              ctypes.POINTER(c_void_p)]),
    'v3d_gemm_free': (None, [c_void_p]),
    'v3d_gemm_gather_f32': (c_int, [c_void_p, c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p),
                                    ctypes.POINTER(c_int), c_int, c_int, c_int, c_float, c_void_p, c_int,
                                    c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    'v3d_sparse_conv_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, ctypes.c_longlong, c_int, c_float, c_void_p,
                                    c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    'v3d_fill_f32': (c_int, [c_void_p, c_size_t, c_float, c_void_p]),
    'v3d_pointnet_input_f32': (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p, c_void_p]),
    'v3d_hash_bytes': (c_size_t, [c_int]),
    'v3d_hash_build': (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'v3d_hash_status': (c_int, [c_void_p, c_int, c_void_p]),
    'v3d_sparse_neighbors': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'v3d_sparse_interp_workspace_bytes': (c_size_t, [c_int, c_int]),
    'v3d_sparse_interp_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int,
                                      c_int, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p, c_size_t,
                                      c_void_p]),
    'v3d_edges_csr_workspace_bytes': (c_size_t, [c_int, c_int]),
    'v3d_edges_csr': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'v3d_edges_csr_status': (c_int, [c_void_p, c_size_t, c_void_p]),
    'v3d_segment_csr_workspace_bytes': (c_size_t, [c_int]),
    'v3d_segment_csr': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'v3d_segment_max_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    'v3d_sort_unique_workspace_bytes': (c_size_t, [c_int]),
    'v3d_sort_unique_u64': (c_int, [c_void_p, c_int, c_void_p, ctypes.POINTER(c_int), c_void_p, c_size_t, c_void_p]),
    'v3d_strided_keys': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'v3d_unpack_coords': (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    'v3d_voxelize_workspace_bytes': (c_size_t, []),
    'v3d_voxel_keys': (c_int, [c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    'v3d_voxelize_status': (c_int, [c_void_p, c_size_t, c_void_p]),
    'v3d_lower_bound_u64': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    'v3d_voxel_decode': (c_int, [c_void_p, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_size_t, c_void_p]),
    'v3d_decoder_fused_f32': (c_int, [ctypes.POINTER(c_void_p), c_void_p, c_void_p, ctypes.POINTER(c_void_p),
                                      ctypes.POINTER(c_int), ctypes.POINTER(c_void_p), ctypes.POINTER(c_int),
                                      ctypes.POINTER(c_int), ctypes.POINTER(c_void_p), ctypes.POINTER(c_float), c_void_p,
                                      c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_size_t, c_void_p]),
    'v3d_decoder_fused_workspace_bytes': (c_size_t, [c_int, c_int]),
    'v3d_conv_pack': (c_int, [c_float_p, c_float_p, c_int, c_int, ctypes.POINTER(c_void_p)]),
    'v3d_conv_free': (None, [c_void_p]),
    'v3d_conv_nhwc_f32': (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p, c_void_p, c_void_p]),
    'v3d_depthwise_nhwc_f32': (c_int, [c_void_p] * 3 + [c_int] * 7 + [c_void_p, c_void_p]),
    'v3d_stem_f32': (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p, c_void_p]),
    'v3d_nhwc_to_nchw_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'v3d_irb_pack': (c_int, [c_float_p] * 6 + [c_int] * 6 + [ctypes.POINTER(c_void_p)]),
    'v3d_irb_free': (None, [c_void_p]),
    'v3d_irb_supported': (c_int, [c_void_p, c_int, c_int]),
    'v3d_irb_workspace_bytes': (c_size_t, [c_void_p, c_int, c_int, c_int]),
    'v3d_stem_block_pack': (c_int, [c_float_p] * 6 + [ctypes.POINTER(c_void_p)]),
    'v3d_stem_block_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'v3d_irb_nhwc_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'v3d_fpn_pack': (c_int, [c_float_p] * 4 + [c_int, ctypes.POINTER(c_void_p)]),
    'v3d_fpn_free': (None, [c_void_p]),
    'v3d_fpn_level_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'v3d_decoder_head_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p]),
}


class V3DLibraryError(RuntimeError):
    """lib3dvnet_hip.so is missing / unloadable, or a call returned an error code."""


_ERR_NAMES = {-1: 'V3D_ERR_BAD_SHAPE', -2: 'V3D_ERR_BAD_ARG', -3: 'V3D_ERR_WORKSPACE_TOO_SMALL',
              -4: 'V3D_ERR_HIP', -5: 'V3D_ERR_UNSUPPORTED'}
_lib = None


def load():
    """Load (once) and return the ctypes handle; raises V3DLibraryError if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise V3DLibraryError('%s not found: the HIP extension is not built (run '
                              '`python 3dvnet_amd/build.py`); there is no CPU fallback' % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise V3DLibraryError('cannot load %s: %s' % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise V3DLibraryError('%s does not export %s' % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    if lib.v3d_version() != ABI_VERSION:
        raise V3DLibraryError('ABI version mismatch: library %d, binding %d'
                              % (lib.v3d_version(), ABI_VERSION))
    _lib = lib
    # developer convenience (scripts/): V3D_OPTIONS="name=value,name=value" -> v3d_set_option once at load time.  (The library
    # itself reads no environment variable.)
    # Every option applied this way is announced on stderr: a stray setting (stop_after, c12_march = 0 ...) changes results
    # or speed of everything the process runs afterwards.
    for item in filter(None, os.environ.get('V3D_OPTIONS', '').split(',')):
        name, _, val = item.partition('=')
        check(lib.v3d_set_option(name.strip().encode(), int(val)), 'v3d_set_option(%s)' % item)
        import warnings
        warnings.warn('3dvnet_amd: developer option %s = %d applied from the V3D_OPTIONS environment variable'
                      % (name.strip(), int(val)), RuntimeWarning, stacklevel=2)
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().v3d_last_error()
        raise V3DLibraryError('%s failed: %s (%s)' % (what, _ERR_NAMES.get(rc, rc),
                                                      msg.decode() if msg else ''))


def set_option(name, value):
    """Developer options of the library (include/v3d.h: v3d_set_option); returns the previous value."""
    lib = load()
    old = c_int(0)
    check(lib.v3d_get_option(name.encode(), ctypes.byref(old)), 'v3d_get_option')
    check(lib.v3d_set_option(name.encode(), int(value)), 'v3d_set_option')
    return old.value


def ptr(t):
    """Device (or host) address of a contiguous tensor, or None."""
    if t is None:
        return None
    assert t.is_contiguous(), 'tensor passed to the C ABI must be contiguous'
    return t.data_ptr()


def stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


def timing_enable(on=True):
    load().v3d_timing_enable(1 if on else 0)


def timing_collect(max_entries=64):
    """-> {kernel name: (total_ms, launches)} since the last collect; synchronises the events."""
    lib = load()
    stride = 64
    names = ctypes.create_string_buffer(max_entries * stride)
    ms = (c_float * max_entries)()
    cnt = (c_int * max_entries)()
    n = lib.v3d_timing_collect(max_entries, names, stride, ms, cnt)
    out = {}
    for i in range(n):
        out[names.raw[i * stride:(i + 1) * stride].split(b'\0')[0].decode()] = (float(ms[i]), int(cnt[i]))
    return out

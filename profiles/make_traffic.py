#!/usr/bin/env python3
"""Derive profiles/rNN_traffic_<cfg>.json (HBM-side bytes per launch of the main kernels) from the two PMC CSVs written by
scripts/profile_bench.sh.  bench.py reads the JSON for `roofline.traffic`.

    python profiles/make_traffic.py <pmc_FETCH_SIZE.csv> <pmc_WRITE_SIZE.csv> <refs_per_step> <out.json>

FETCH_SIZE correction (MI355X_MICROARCH.md §HBM): on gfx950 rocprofv3's FETCH_SIZE reports exactly half of the bytes of a
wide coalesced read (16 B per lane; the counter tallies 128-byte requests as 64 B).  EVERY kernel listed here reads its
bulk data with 16-byte-per-lane loads (the warp kernel gathers 128-byte cells as 8 lanes x 16 B, conv0 / convg / deconvg /
conv9+prob copy 16-byte slots, soft_argmin is the exception with 4-byte loads and negligible traffic), so the x2 is applied
uniformly: hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Cross-check from the data itself: conv1's raw FETCH_SIZE is
0.30 GB per launch while it must read conv0's whole 0.62 GB output exactly once.  WRITE_SIZE needs no correction
(the warp kernel's WRITE_SIZE equals its output volume to the byte).
"""
import csv
import json
import sys

# substring of the (shortened) kernel symbol -> the name bench.py / the library's timing log uses
NAMES = [('psv_variance_window_kernel<true, false>', 'psv_variance'), ('psv_variance_window_kernel<true, true>', 'psv_variance_cl8'),
         ('psv_variance_window_kernel<true>', 'psv_variance'), ('psv_variance_reuse_kernel<true>', 'psv_variance'),
         ('psv_variance_window_kernel', 'psv_variance_f32_out'), ('psv_variance_reuse_kernel', 'psv_variance_f32_out'), ('psv_variance_kernel', 'psv_variance'),
         ('conv0z_kernel<false>', 'costreg_conv0'), ('conv0z_kernel<true>', 'costreg_conv0_f32'), ('conv0_bf16x2_kernel', 'costreg_conv0'), ('conv12z_kernel', 'costreg_conv12'), ('conv9_prob_kernel<true>', 'costreg_conv9_prob_f32'), ('conv9_prob_kernel', 'costreg_conv9_prob'),
         ('convg_bf16x2_kernel<CG<8, 16', 'costreg_conv1'), ('convg_bf16x2_kernel<CG<16, 16', 'costreg_conv2'),
         ('convg_bf16x2_kernel<CG<16, 32', 'costreg_conv3'), ('convg_bf16x2_kernel<CG<32, 32', 'costreg_conv4'),
         ('deconvg_bf16x2_kernel<DG<32, 16', 'costreg_conv8'), ('soft_argmin_kernel', 'soft_argmin'),
         ('decoder_fused_kernel', 'decoder_fused'), ('decoder_corner_kernel', 'decoder_corners'),
         ('gemm_gather_rounds_kernel<2, 2, 4>', 'sparse_conv_gemm'),
         ('backproject_variance_kernel', 'backproject_variance')]
FETCH_CORRECTION = 2.0
# kernels whose bulk reads are 4-byte-per-lane loads: FETCH_SIZE is already the byte count (the x2 applies to 16-byte-per-lane reads
# only).  propz_kernel's loaders read the guide features / image one float per lane (cross-check: its WRITE_SIZE is the output to the
# byte, its raw FETCH_SIZE 1.3-1.4x the guide + depth bytes -- the 48 / 40 halo columns and the rows the strips share).
NO_CORRECTION = {'propagation_fused'}


def read(path):
    out = {}
    for row in csv.DictReader(open(path)):
        for sub, name in NAMES:
            if sub in row['kernel']:
                out[name] = float(row['avg_value_per_dispatch'])
                break
    return out


def weighted(path, match):
    """dispatch-weighted mean of the counter over the kernels whose symbol satisfies `match`, and their dispatch count"""
    tot, n = 0.0, 0
    for row in csv.DictReader(open(path)):
        if match(row['kernel']):
            tot += float(row['avg_value_per_dispatch']) * int(row['dispatches'])
            n += int(row['dispatches'])
    return (tot / n if n else 0.0), n


# kernel families with several template instances per scene: the sparse convolutions of the U-Net (pipeline kernel, 32- / 64-row
# tiles, 64 / 128 channels) and the twelve conv launches of stage 3 (FLAT instances of convg_bf16x2_kernel: "..., true> >")
FAMILIES = [('sparse_conv_gemm', lambda k: 'gemm_gather_pipe_kernel' in k),
            ('propagation_conv', lambda k: 'convg_bf16x2_kernel' in k and 'deconvg' not in k and k.rstrip().endswith('true> >')),
            # round 6: one row-marching kernel per PropagationNet (csrc/propz.hip), three instances per scene
            ('propagation_fused', lambda k: 'propz_kernel' in k and 'false>' in k)]


def main():
    fetch, write = read(sys.argv[1]), read(sys.argv[2])
    for name, match in FAMILIES:
        (f, nf), (w, nw) = weighted(sys.argv[1], match), weighted(sys.argv[2], match)
        if nf and nw:
            fetch[name], write[name] = f, w
    kernels = {}
    for k in sorted(fetch.keys() | write.keys()):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        corr = 1.0 if k in NO_CORRECTION else FETCH_CORRECTION
        kernels[k] = {'fetch_kb_raw': f, 'write_kb': w, 'fetch_bytes_corrected': corr * f * 1024.0,
                      'write_bytes': w * 1024.0, 'hbm_bytes': (corr * f + w) * 1024.0, 'fetch_correction': corr}
    json.dump({'refs_per_step_per_gpu': int(sys.argv[3]),
               'unit': 'per launch; fetch_kb_raw / write_kb = rocprofv3 FETCH_SIZE / WRITE_SIZE (KB); hbm_bytes = '
                       '(2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 correction for 16-byte-per-lane reads applied '
                       'to every kernel that reads that way (fetch_correction; profiles/README.md)',
               'kernels': kernels}, open(sys.argv[4], 'w'), indent=1)


if __name__ == '__main__':
    main()

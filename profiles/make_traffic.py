#!/usr/bin/env python3
"""Derive profiles/rNN_traffic.json (HBM-side FETCH_SIZE + WRITE_SIZE per launch of the main kernels) from
the two PMC CSVs written by scripts/profile_bench.sh.  bench.py reads the JSON for `roofline.traffic`.

    python profiles/make_traffic.py <pmc_FETCH_SIZE.csv> <pmc_WRITE_SIZE.csv> <refs_per_step> <out.json>
"""
import csv
import json
import sys

# substring of the (shortened) kernel symbol -> the name bench.py / the library's timing log uses
NAMES = [('psv_variance_reuse_kernel', 'psv_variance'), ('psv_variance_kernel', 'psv_variance'),
         ('conv0_bf16x2_kernel', 'costreg_conv0'), ('conv9_prob_kernel', 'costreg_conv9_prob'),
         ('convg_bf16x2_kernel<CG<8, 16', 'costreg_conv1'), ('convg_bf16x2_kernel<CG<16, 16', 'costreg_conv2'),
         ('deconvg_bf16x2_kernel<DG<32, 16', 'costreg_conv8'), ('soft_argmin_kernel', 'soft_argmin')]


def read(path):
    out = {}
    for row in csv.DictReader(open(path)):
        for sub, name in NAMES:
            if sub in row['kernel']:
                out[name] = float(row['avg_value_per_dispatch'])
    return out


def main():
    fetch, write = read(sys.argv[1]), read(sys.argv[2])
    kernels = {k: {'fetch_kb': fetch.get(k, 0.0), 'write_kb': write.get(k, 0.0)} for k in fetch.keys() | write.keys()}
    json.dump({'refs_per_step_per_gpu': int(sys.argv[3]),
               'unit': 'KB per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE, gfx950, uncorrected; see profiles/README.md)',
               'kernels': kernels}, open(sys.argv[4], 'w'), indent=1)


if __name__ == '__main__':
    main()

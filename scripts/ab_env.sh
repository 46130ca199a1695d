#!/bin/bash
# Developer A/B of runtime switches on one box: alternates `bench_layers.py` between environment settings (each argument one
# setting, "" = default), N rounds, prints every run; e.g.  bash scripts/ab_env.sh 4 "" "V3D_OPTIONS=psv_kernel=1"
# (V3D_OPTIONS: developer options of the library, applied by 3dvnet_amd/_lib.py through v3d_set_option)
N=$1; shift
for r in $(seq 1 $N); do
  for e in "$@"; do
    env $e python scripts/bench_layers.py --refs 64 --iters 20 --tag "[$e]" 2>&1 | grep total | cut -c1-150
  done
done

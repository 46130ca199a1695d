#!/bin/bash
# Developer: phase shares and variants of the depth-march conv0 (variants prebuilt with scripts/build_variant.py):
#   bash scripts/conv0z_ab.sh czm1 czh1 ...
A=3dvnet_amd/build/ablate
[ -f $A/lib_czph.so ] && V3D_LIB_OVERRIDE=$A/lib_czph.so python scripts/phase_conv0z.py
for r in 1 2; do
  python scripts/bench_layers.py --refs 64 --iters 20 --tag "[default]" | cut -c1-200
  for v in "$@"; do
    V3D_LIB_OVERRIDE=$A/lib_$v.so python scripts/bench_layers.py --refs 64 --iters 20 --tag "[$v]" | cut -c1-200
  done
done

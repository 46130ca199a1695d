#!/bin/bash
# Developer A/B on the GPU box: for every flag set given as an argument (quote each set), rebuild the library with
# V3D_EXTRA_FLAGS=<set> (objects are keyed by flags, so variants do not clobber each other) and print the per-kernel times
# of the cfg2 path.   bash scripts/ab_build.sh "" "-DV3D_PSV_RDB=8 -DV3D_PSV_WAVES=4"
for f in "$@"; do
  V3D_EXTRA_FLAGS="$f" python 3dvnet_amd/build.py > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  python scripts/bench_layers.py --refs 64 --iters 10 --tag "[$f]" 2>&1 | grep total | cut -c1-260
done
python 3dvnet_amd/build.py > /dev/null 2>&1     # restore the default build

set -e
run() { # tag, flags
  V3D_EXTRA_FLAGS="$2" python 3dvnet_amd/build.py --force > /dev/null 2>&1 && python scripts/bench_layers.py --tag "$1" $3 2>&1 | tail -1 | cut -c1-130
}
run "base" ""
run "no_mfma" "-DV3D_ABLATE=1"
run "no_restage" "-DV3D_ABLATE=2"

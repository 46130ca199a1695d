#!/bin/bash
# Developer: marginal costs of the fused hypothesis decoder.  Part 1 (`build`, runs anywhere hipcc does): decoder.hip compiled
# once per -DV3D_FUSED_ABLATE=n (and any extra flag sets given) and linked with the default objects of every other source into
# 3dvnet_amd/build/ablate/lib_<tag>.so.  Part 2 (`run`, on the GPU box): every variant timed with fused_decoder_time.py
# (V3D_LIB_OVERRIDE picks the library), default library last.
#   bash scripts/micro/fused_decoder_ablate.sh build ["-DFLAG ..." ...];  bash scripts/micro/fused_decoder_ablate.sh run
set -e
cd "$(dirname "$0")/../.."
mode=$1; shift || true
out=3dvnet_amd/build/ablate
mkdir -p $out
if [ "$mode" = build ]; then
  python 3dvnet_amd/build.py > /dev/null
  def=3dvnet_amd/build/$(cat 3dvnet_amd/build/linked_flags)
  others=$(ls $def/*.o | grep -v /decoder.o)
  sets=("$@")
  if [ ${#sets[@]} -eq 0 ]; then sets=("-DV3D_FUSED_ABLATE=1" "-DV3D_FUSED_ABLATE=2" "-DV3D_FUSED_ABLATE=3" "-DV3D_FUSED_ABLATE=4" "-DV3D_FUSED_ABLATE=5" "-DV3D_FUSED_ABLATE=6"); fi
  for f in "${sets[@]}"; do
    tag=$(echo "$f" | tr -c 'A-Za-z0-9=\n' '_')
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I3dvnet_amd/csrc $f -c 3dvnet_amd/csrc/decoder.hip -o $out/dec_$tag.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/lib_$tag.so $others $out/dec_$tag.o && echo "built $tag" ) &
  done
  wait
else
  for lib in $out/lib_*.so; do
    echo "== $lib"
    V3D_LIB_OVERRIDE=$PWD/$lib V3D_TIME_FUSED_ONLY=1 python scripts/micro/fused_decoder_time.py 2>&1 | grep "fused=True\|phases" | cut -c1-400
  done
  echo "== default"
  V3D_TIME_FUSED_ONLY=1 python scripts/micro/fused_decoder_time.py 2>&1 | grep "fused=True\|phases" | cut -c1-400
fi

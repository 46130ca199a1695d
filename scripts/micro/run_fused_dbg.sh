# developer: stress and timing of the fused hypothesis decoder, 16 and 8 query points per workgroup
for f in "-DV3D_FUSED_PTS=8" "-DV3D_FUSED_PTS=16"; do
  V3D_EXTRA_FLAGS="$f" python 3dvnet_amd/build.py > /tmp/b.log 2>&1 || { echo "build failed: $f"; tail -3 /tmp/b.log; continue; }
  echo "== $f"
  python scripts/micro/fused_decoder_stress.py --reps 100 --views 8 --noise 2>&1 | grep -E "LDS_KB"
  python scripts/micro/fused_decoder_time.py 2>&1 | grep -E "LDS|max"
done
python 3dvnet_amd/build.py > /dev/null 2>&1

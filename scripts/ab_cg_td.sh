# developer A/B of convg variants built with scripts/build_variant.py: bash scripts/ab_cg_td.sh tag1 tag2 ...
cd /root/repo
for t in base "$@" base; do
  if [ $t = base ]; then unset V3D_LIB_OVERRIDE; else export V3D_LIB_OVERRIDE=3dvnet_amd/build/ablate/lib_$t.so; fi
  python scripts/bench_layers.py --refs 64 --tag $t 2>&1 | tail -1 | cut -c1-600
done

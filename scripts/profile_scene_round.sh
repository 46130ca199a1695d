#!/bin/bash
# The scene half of a round's profiles: kernel stats of `bench.py --config cfg3 --stage3` (13 scenes) and the PMC passes of
# scripts/profile_scene_pmc.sh with the traffic JSON -> gpurun_out/profile_cfg3full, gpurun_out/profile_cfg3
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/profile_cfg3full; T=/tmp/v3dprof_cfg3full; rm -rf $T; mkdir -p $O $T; cd /tmp
rocprofv3 --kernel-trace --stats -d $T/kt -o r -- python $R/bench.py --config cfg3 --stage3 --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_under_rocprof.log 2>&1
python $R/profiles/summarize_rocpd.py stats $T/kt/r_results.db $O/kernel_stats.csv
cd $R
timeout 900 bash scripts/profile_scene_pmc.sh > gpurun_out/r5_profile_scene_pmc.log 2>&1
python profiles/make_traffic.py gpurun_out/profile_cfg3/pmc_FETCH_SIZE.csv gpurun_out/profile_cfg3/pmc_WRITE_SIZE.csv 64 gpurun_out/profile_cfg3/traffic.json
head -14 $O/kernel_stats.csv | cut -c1-140

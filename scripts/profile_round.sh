#!/bin/bash
# Round-6 profiles: rocprofv3 kernel stats + PMC passes of the bench command (cfg2, 64 views; cfg5, 8 views) and the stage-3 /
# scene kernel stats.  Summaries land in gpurun_out/profile_*; the ones to be judged are copied to profiles/r06_*.
bash scripts/profile_bench.sh cfg2 64 > gpurun_out/r6_profile_cfg2.log 2>&1
bash scripts/profile_bench.sh cfg5 8 > gpurun_out/r6_profile_cfg5.log 2>&1
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/profile_cfg3full; T=/tmp/v3dprof_cfg3full; rm -rf $T; mkdir -p $O $T; cd /tmp
rocprofv3 --kernel-trace --stats -d $T/kt -o r -- python $R/bench.py --config cfg3 --stage3 --no-cpu-baseline --no-fp32 --steps 10 --warmup 2 > $O/bench_under_rocprof.log 2>&1
python $R/profiles/summarize_rocpd.py stats $T/kt/r_results.db $O/kernel_stats.csv
cd $R; tail -2 gpurun_out/r6_profile_cfg2.log; head -25 gpurun_out/profile_cfg2/kernel_stats.csv | cut -c1-150
bash scripts/profile_scene_pmc.sh > gpurun_out/r6_profile_cfg3_pmc.log 2>&1
python profiles/make_traffic.py gpurun_out/profile_cfg3/pmc_FETCH_SIZE.csv gpurun_out/profile_cfg3/pmc_WRITE_SIZE.csv 64 gpurun_out/profile_cfg3/traffic.json

#!/bin/bash
# Kernel-trace statistics of the full-pipeline bench (`bench.py --config cfg3`): every kernel of a scene, the library's and
# torch's, summed per name -> gpurun_out/profile_cfg3/kernel_stats.csv (copy into profiles/ as rNN_kernel_stats_cfg3.csv).
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/profile_cfg3; T=/tmp/v3dprof_cfg3; rm -rf $T; mkdir -p $O $T; cd /tmp
STEPS=${1:-10}
rocprofv3 --kernel-trace --stats -d $T/kt -o r -- python $R/bench.py --config cfg3 --no-cpu-baseline --no-extra --steps $STEPS --warmup 2 > $O/bench_under_rocprof.log 2>&1
python $R/profiles/summarize_rocpd.py stats $T/kt/r_results.db $O/kernel_stats.csv
tail -1 $O/bench_under_rocprof.log | cut -c1-300
head -45 $O/kernel_stats.csv | cut -c1-150

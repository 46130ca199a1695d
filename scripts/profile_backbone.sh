#!/bin/bash
# rocprofv3 kernel stats of the backbone leg of bench.py (native kernels, then the stock torch modules for comparison) -> gpurun_out/profile_backbone
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/profile_backbone; T=/tmp/v3dprof_bb; rm -rf $T; mkdir -p $O $T; cd /tmp
rocprofv3 --kernel-trace --stats -d $T/kt -o r -- bash -c "cd $R && bash scripts/micro/backbone_total.sh" > $O/bench_under_rocprof.log 2>&1
python $R/profiles/summarize_rocpd.py stats $T/kt/r_results.db $O/kernel_stats.csv
head -30 $O/kernel_stats.csv | cut -c1-150

#!/bin/bash
# Developer: GPU idle time inside the full-pipeline bench (`bench.py --config cfg3`, 3 timed scenes): which kernels the
# device waits for (gaps mode of profiles/summarize_rocpd.py over the dispatches of the last scene).
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/trace_scene; T=/tmp/v3dtrace_scene; rm -rf $T; mkdir -p $O $T; cd /tmp
rocprofv3 --kernel-trace -d $T/kt -o r -- python $R/bench.py --config cfg3 --no-cpu-baseline --no-extra --steps 3 --warmup 1 > $O/log.txt 2>&1
python $R/profiles/summarize_rocpd.py gaps $T/kt/r_results.db $O/gaps.csv ${1:-1100}
python $R/profiles/summarize_rocpd.py trace $T/kt/r_results.db $O/trace.csv ${1:-1100}
cat $O/gaps.csv | cut -c1-150

#!/bin/bash
# Developer: the kernels of one SparseUNet.forward in launch order (rocprofv3 kernel trace of scripts/unet_host_profile.py).
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/trace_unet; T=/tmp/v3dtrace_unet; rm -rf $T; mkdir -p $O $T; cd /tmp
rocprofv3 --kernel-trace -d $T/kt -o r -- python $R/scripts/unet_host_profile.py > $O/log.txt 2>&1
python - <<PY
import sqlite3
cur = sqlite3.connect('$T/kt/r_results.db').cursor()
print([d[0] for d in cur.execute("select * from kernels limit 1").description])
PY
python $R/profiles/summarize_rocpd.py trace $T/kt/r_results.db $O/trace.csv ${1:-110} || true
cat $O/trace.csv | cut -c1-140

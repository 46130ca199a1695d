#!/bin/bash
# round-3 GPU call 1: masked-gather micro-benchmark, warp-kernel ablations, counter list
mkdir -p gpurun_out/r3c1
timeout 300 scripts/micro/_bin/ta_mask > gpurun_out/r3c1/ta_mask.txt 2>&1
timeout 1500 bash scripts/ab_build.sh "" "-DV3D_PSV_ABLATE=1" "-DV3D_PSV_ABLATE=2" "-DV3D_PSV_ABLATE=3" "-DV3D_PSV_ABLATE=4" "-DV3D_PSV_NOSERIAL" "" > gpurun_out/r3c1/ablate.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 -L > /root/repo/gpurun_out/r3c1/counters_all.txt 2>&1)
grep -E "TA_|TCP_|TD_" gpurun_out/r3c1/counters_all.txt | cut -c1-200 | head -150 > gpurun_out/r3c1/counters_ta.txt
cat gpurun_out/r3c1/ta_mask.txt gpurun_out/r3c1/ablate.txt

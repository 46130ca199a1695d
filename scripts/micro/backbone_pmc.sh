#!/bin/bash
# Developer: SQ counters of the native backbone's kernels (one --pmc pass over scripts/micro/backbone_total.sh)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/profile_backbone; T=/tmp/v3dprof_bbpmc; rm -rf $T; mkdir -p $O $T; cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_MFMA -d $T/sq -o r -- bash -c "cd $R && bash scripts/micro/backbone_total.sh" > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py pmc $T/sq/r_results.db $O/pmc_sq.csv
grep -h "conv_gemm_kernel<1, 9, false>\|conv_gemm_kernel<4, 1, false>\|conv_gemm_kernel<2, 1, true>" $O/pmc_sq.csv | cut -c1-110

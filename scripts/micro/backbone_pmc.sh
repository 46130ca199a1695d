#!/bin/bash
# Developer: PMC passes of the native backbone's kernels (each --pmc pass its own run over scripts/micro/backbone_total.sh):
# SQ matrix / wait counters, LDS / VALU counters, FETCH_SIZE / WRITE_SIZE -> gpurun_out/profile_backbone/pmc_*.csv
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/profile_backbone; T=/tmp/v3dprof_bbpmc; rm -rf $T; mkdir -p $O $T; cd /tmp
B="bash -c \"cd $R && bash scripts/micro/backbone_total.sh\""
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_MFMA -d $T/sq -o r -- bash -c "cd $R && bash scripts/micro/backbone_total.sh" > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py pmc $T/sq/r_results.db $O/pmc_sq.csv
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -d $T/sq2 -o r -- bash -c "cd $R && bash scripts/micro/backbone_total.sh" > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py pmc $T/sq2/r_results.db $O/pmc_sq2.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $T/$c -o r -- bash -c "cd $R && bash scripts/micro/backbone_total.sh" > /dev/null 2>&1
  python $R/profiles/summarize_rocpd.py pmc $T/$c/r_results.db $O/pmc_$c.csv
done
grep -h "irb_kernel\|fpn_level" $O/pmc_sq.csv | cut -c1-160 | head -20

#!/bin/bash
# PMC passes over one point-flow sweep (scripts/micro/fused_decoder_time.py): SQ counters of the fused decoder -> gpurun_out/pmc_decoder/
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/pmc_decoder; T=/tmp/v3dpmc_dec; rm -rf $T; mkdir -p $O $T; cd /tmp
B="python $R/scripts/micro/fused_decoder_time.py"
export V3D_TIME_FUSED_ONLY=1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_MFMA -d $T/sq -o r -- $B > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py pmc $T/sq/r_results.db $O/pmc_sq.csv
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU -d $T/sq2 -o r -- $B > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py pmc $T/sq2/r_results.db $O/pmc_sq2.csv
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM -d $T/sq3 -o r -- $B > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py pmc $T/sq3/r_results.db $O/pmc_sq3.csv
head -1 $O/pmc_sq.csv; grep -h "decoder_fused" $O/pmc_sq.csv; head -1 $O/pmc_sq2.csv; grep -h "decoder_fused" $O/pmc_sq2.csv; head -1 $O/pmc_sq3.csv; grep -h "decoder_fused" $O/pmc_sq3.csv

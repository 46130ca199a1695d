#!/bin/bash
# PMC passes over the full-pipeline scene bench (`bench.py --config cfg3 --stage3`): FETCH_SIZE / WRITE_SIZE and the SQ counters of
# the fused hypothesis decoder and the sparse-conv GEMMs (each --pmc pass its own run) -> gpurun_out/profile_cfg3/.
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/profile_cfg3; T=/tmp/v3dprof_cfg3; rm -rf $T; mkdir -p $O $T; cd /tmp
B="python $R/bench.py --config cfg3 --stage3 --no-cpu-baseline --no-fp32 --steps 3 --warmup 1"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $T/$c -o r -- $B > /dev/null 2>&1
  python $R/profiles/summarize_rocpd.py pmc $T/$c/r_results.db $O/pmc_$c.csv
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_MFMA -d $T/sq -o r -- $B > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py pmc $T/sq/r_results.db $O/pmc_sq.csv
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -d $T/sq2 -o r -- $B > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py pmc $T/sq2/r_results.db $O/pmc_sq2.csv
grep -h "decoder_fused" $O/pmc_*.csv | cut -c1-120

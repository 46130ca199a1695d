#!/bin/bash
# PMC passes over the cfg2 cost-volume path (scripts/bench_layers.py, 64 views): SQ issue / wait / LDS counters and -- in their
# own bounded runs, they have hung on this pool before -- the TA / TCP counters.  $1 = output tag, remaining args = env settings.
TAG=$1; shift
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/pmc_$TAG; T=/tmp/pmc_$TAG; rm -rf $T; mkdir -p $O $T; cd /tmp
B="env $@ python $R/scripts/bench_layers.py --refs 64 --iters 3"
pass() {  # name, counters...
  n=$1; shift
  timeout 240 rocprofv3 --kernel-trace --pmc "$@" -d $T/$n -o r -- $B > $O/$n.log 2>&1
  rc=$?
  if [ $rc -eq 0 ] && [ -f $T/$n/r_results.db ]; then python $R/profiles/summarize_rocpd.py pmc $T/$n/r_results.db $O/pmc_$n.csv; else echo "pass $n failed rc=$rc" | tee -a $O/failed.txt; fi
}
pass sq SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
pass sq3 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL
pass ta TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES
pass tcp TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES
grep -h "psv_variance" $O/pmc_*.csv | sort -t, -k2 | cut -c1-140

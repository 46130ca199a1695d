#!/bin/bash
# Profile `bench.py --config $1` (default cfg2) with rocprofv3 on the GPU box and leave only small CSV summaries in
# gpurun_out/profile_$1/ (kernel-trace stats, FETCH_SIZE / WRITE_SIZE and SQ counter passes -- PMC passes are separate runs,
# never combined with other trace domains).  Copy what should be judged into profiles/ as rNN_*.
CFG=${1:-cfg2}; REFS=${2:-64}
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/profile_$CFG; T=/tmp/v3dprof_$CFG; rm -rf $T; mkdir -p $O $T; cd /tmp
B="python $R/bench.py --config $CFG --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats -d $T/kt -o r -- $B --steps 10 --warmup 2 > $O/bench_under_rocprof.log 2>&1
python $R/profiles/summarize_rocpd.py stats $T/kt/r_results.db $O/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $T/$c -o r -- $B --steps 3 --warmup 1 > /dev/null 2>&1
  python $R/profiles/summarize_rocpd.py pmc $T/$c/r_results.db $O/pmc_$c.csv
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_MFMA -d $T/sq -o r -- $B --steps 3 --warmup 1 > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py pmc $T/sq/r_results.db $O/pmc_sq.csv
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $T/sq2 -o r -- $B --steps 3 --warmup 1 > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py pmc $T/sq2/r_results.db $O/pmc_sq2.csv
# (a pass with the TA_* counters -- TA_BUSY_avr, TA_FLAT_READ_WAVEFRONTS_sum, ... -- never returned on this pool: not collected)
tail -1 $O/bench_under_rocprof.log | cut -c1-200
python $R/profiles/make_traffic.py $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv $REFS $O/traffic.json
ls -la $O

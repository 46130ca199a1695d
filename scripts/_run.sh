export TMPDIR=/tmp
R=$PWD; T=/tmp/pzpmc; rm -rf $T; mkdir -p $T gpurun_out; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $T/a -o r -- python $R/scripts/bench_stage3_nets.py > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py pmc $T/a/r_results.db $R/gpurun_out/r6_propz_pmc_planes.csv
grep propz $R/gpurun_out/r6_propz_pmc_planes.csv

#!/bin/bash
# LDS bank-conflict counters of the bench kernels: gpurun_out/pmc_lds.csv
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; T=/tmp/v3dpmc; rm -rf $T; mkdir -p $O $T; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d $T/l -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${@} > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py pmc $T/l/r_results.db $O/pmc_lds.csv
grep -E "psv_variance|conv0_bf16|conv9_prob|gemm_gather" $O/pmc_lds.csv

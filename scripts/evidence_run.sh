#!/bin/bash
# Round-6 evidence run on the GPU box: the whole GPU suite, smoke, the default bench line (N = 1)
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r6_pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_smoke.txt 2>&1
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r6_bench_n1.json 2> gpurun_out/r6_bench_n1.err ) 2> gpurun_out/r6_bench_time.txt
cat gpurun_out/r6_pytest_gpu_tail.txt gpurun_out/r6_smoke.txt gpurun_out/r6_bench_time.txt

#!/bin/bash
# end-of-round evidence: GPU suite, the three bench lines (un-profiled), parity report
mkdir -p gpurun_out/r3final
timeout 2700 python -m pytest tests -m gpu -x -q > gpurun_out/r3final/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r3final/pytest_gpu.txt
timeout 1200 python bench.py > gpurun_out/r3final/bench_n1.json 2> gpurun_out/r3final/bench_n1.err; tail -c 600 gpurun_out/r3final/bench_n1.json
timeout 900 python bench.py --config cfg5 --check-refs 2 --host-check-refs 1 > gpurun_out/r3final/bench_cfg5.json 2>/dev/null
timeout 900 python bench.py --config cfg3 > gpurun_out/r3final/bench_cfg3.json 2>/dev/null
timeout 900 python scripts/parity_report.py cfg1 cfg2 cfg5 > gpurun_out/r3final/parity_report.txt 2>&1
bash scripts/profile_scene.sh 10 > gpurun_out/r3final/profile_scene.txt 2>&1; cp gpurun_out/profile_cfg3/kernel_stats.csv gpurun_out/r3final/kernel_stats_cfg3.csv
bash scripts/profile_bench.sh cfg2 64 > gpurun_out/r3final/profile_cfg2.txt 2>&1

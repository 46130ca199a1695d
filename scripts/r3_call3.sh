#!/bin/bash
mkdir -p gpurun_out/r3c3
python scripts/bench_layers.py --refs 64 --iters 10 --tag window 2>&1 | grep total | cut -c1-300
V3D_PSV_REUSE=1 python scripts/bench_layers.py --refs 64 --iters 10 --tag reuse 2>&1 | grep total | cut -c1-300
timeout 1500 python -m pytest tests/test_costvolume_gpu.py tests/test_parity_net_gpu.py -m gpu -x -q 2>&1 | tail -15
python scripts/bench_layers.py --refs 64 --iters 10 --tag window 2>&1 | grep total | cut -c1-300

timeout 900 python -m pytest tests/test_parity_net_gpu.py -x -q -m gpu -k "propagation or upsampling" 2>&1 | tail -8
python scripts/bench_stage3_nets.py --tag split
python scripts/bench_stage3_nets.py --tag fp32 --precision fp32

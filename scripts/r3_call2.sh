#!/bin/bash
mkdir -p gpurun_out/r3c2
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3c2/pytest.txt 2>&1
tail -15 gpurun_out/r3c2/pytest.txt
( time timeout 1500 python bench.py ) > gpurun_out/r3c2/bench.txt 2> gpurun_out/r3c2/bench.err
tail -c 3000 gpurun_out/r3c2/bench.txt; tail -5 gpurun_out/r3c2/bench.err

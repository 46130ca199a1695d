#!/bin/bash
timeout 900 python -m pytest tests/test_costvolume_gpu.py tests/test_scene_gpu.py -m gpu -x -q 2>&1 | tail -3
bash scripts/ab_build.sh "" "-DV3D_PSV_WAVES=3" "-DV3D_PSV_WAVES=2" "" 2>&1 | cut -c1-130

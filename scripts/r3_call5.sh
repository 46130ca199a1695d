#!/bin/bash
bash scripts/ab_build.sh "" "-DV3D_PSV_NOPIPE" "-DV3D_PSV_WAVES=3" "-DV3D_PSV_WAVES=3 -DV3D_PSV_NOPIPE" "" 2>&1 | cut -c1-130
python 3dvnet_amd/build.py >/dev/null 2>&1; timeout 900 python -m pytest tests/test_costvolume_gpu.py -m gpu -x -q 2>&1 | tail -3

#!/bin/bash
mkdir -p gpurun_out/r3c4
bash scripts/ab_build.sh "" "-DV3D_PSVW_ABLATE=1" "-DV3D_PSVW_ABLATE=2" "-DV3D_PSVW_ABLATE=3" "-DV3D_PSVW_ABLATE=4" "-DV3D_PSVW_ABLATE=5" "-DV3D_PSVW_ABLATE=6" "" 2>&1 | cut -c1-120 | tee gpurun_out/r3c4/ablate_window.txt

#!/bin/bash
# run on the GPU box: rebuild with phase instrumentation into the in-tree .so, run, then the caller rebuilds normally
V3D_EXTRA_FLAGS="-DV3D_PHASE_TIMING" python 3dvnet_amd/build.py --force > /dev/null 2>&1
python scripts/phase_timing.py "$@"

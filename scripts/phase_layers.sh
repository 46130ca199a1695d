#!/bin/bash
V3D_EXTRA_FLAGS="-DV3D_PHASE_TIMING" python 3dvnet_amd/build.py --force > /dev/null 2>&1
for l in "$@"; do python scripts/phase_layer.py --layer $l 2>&1 | tail -1; done

#!/bin/bash
# kernel resource usage of one csrc file: scripts/kres.sh costreg [grep pattern]
cd /root/repo/3dvnet_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../include -I . $V3D_EXTRA_FLAGS -c $1.hip -o /tmp/kres_$1.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - - - - | cut -c1-220 | grep -E "${2:-.}"

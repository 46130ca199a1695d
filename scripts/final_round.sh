#!/bin/bash
# Round-6 final evidence on the GPU box: profiles (scripts/profile_round.sh), then the GPU suite, smoke and the default bench line
bash scripts/profile_round.sh > gpurun_out/r6_profile_round.log 2>&1
bash scripts/evidence_run.sh

#!/bin/bash
# Step time of the headline batch against the chunk length (through gpurun): bash tools/chunk_sweep.sh [lengths...]
cd "$(dirname "$0")/.."
for c in ${@:-12 16 20 25 34 50}; do echo chunk $c; PDP_SOLVE_CHUNK=$c timeout 300 python tools/ab_probe.py libpdp_hip.so libpdp_hip.so 1 2>&1 | grep -v amdgpu.ids; done

for c in 12 14 16 18 20 25; do echo chunk $c; PDP_SOLVE_CHUNK=$c timeout 300 python tools/ab_probe.py libpdp_hip.so libpdp_hip.so 1 2>&1 | grep -v amdgpu.ids; done

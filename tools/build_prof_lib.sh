#!/bin/bash
# Builds pdp-solver_amd/csrc/libpdp_hip_prof.so: the library with -DPDP_PHASE_PROF (register-accumulated cycle counters per solver phase,
# read by tools/phase_prof.py) next to the product library, from a scratch copy of the sources.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/pdp_profbuild; rm -rf $B; mkdir -p $B
cp $ROOT/pdp-solver_amd/csrc/*.hip $ROOT/pdp-solver_amd/csrc/*.hpp $ROOT/pdp-solver_amd/csrc/Makefile $B/
sed -i "s#\.\./\.\./include#$ROOT/include#g" $B/Makefile $B/*.hpp
make -C $B -j6 EXTRA=-DPDP_PHASE_PROF 2>&1 | grep -E "error|Error" || true
cp $B/libpdp_hip.so $ROOT/pdp-solver_amd/csrc/libpdp_hip_prof.so
cp $B/libpdp_hip_fast.so $ROOT/pdp-solver_amd/csrc/libpdp_hip_fast_prof.so        # PDP_BUILD=fast python tools/phase_prof.py
ls -la $ROOT/pdp-solver_amd/csrc/libpdp_hip_prof.so $ROOT/pdp-solver_amd/csrc/libpdp_hip_fast_prof.so

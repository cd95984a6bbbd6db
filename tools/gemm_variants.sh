#!/bin/bash
# builds variants of the library that differ in pdp_train.hip's compile-time knobs: tools/gemm_variants.sh name "-DX=1 ..." [name flags ...]
cd "$(dirname "$0")/../pdp-solver_amd/csrc"
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc $2 -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -Wno-pass-failed -c pdp_train.hip -o /tmp/pdp_train_$1.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o libv_$1.so pdp_problem.o pdp_ops.o pdp_walksat.o pdp_solve.o pdp_neural.o /tmp/pdp_train_$1.o pdp_dimacs.o pdp_coo.o || exit 1
  shift 2
done

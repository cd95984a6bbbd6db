#!/bin/bash
# A/B builds: tools/build_variant_lib.sh NAME "-DFLAG ..." [file.hip ...] compiles the named sources (default pdp_neural.hip) with the extra flags
# and links them with the product's other objects into pdp-solver_amd/csrc/libpdp_hip_NAME.so (select it with PDP_HIP_LIB on the GPU box).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/pdp-solver_amd/csrc
NAME=$1; EXTRA=$2; shift 2 || true
FILES=${@:-pdp_neural.hip}
B=/tmp/pdp_variant_$NAME; rm -rf $B; mkdir -p $B
OBJS=""
for f in pdp_problem pdp_ops pdp_walksat pdp_solve pdp_neural pdp_train pdp_dimacs pdp_coo; do
    if echo " $FILES " | grep -q " $f.hip "; then
        /opt/rocm/bin/hipcc $EXTRA -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-result -Wno-pass-failed -I$C -c $C/$f.hip -o $B/$f.o
        OBJS="$OBJS $B/$f.o"
    else
        OBJS="$OBJS $C/$f.o"
    fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $C/libpdp_hip_$NAME.so $OBJS
ls -la $C/libpdp_hip_$NAME.so

#!/bin/bash
# Builds pdp-solver_amd/csrc/libpdp_hip_prev.so from the sources of a git revision (default HEAD): the "before" side of a same-box A/B run
# through PDP_HIP_LIB (the working tree's libpdp_hip.so is the "after" side).
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/pdp_prevbuild; rm -rf $B; mkdir -p $B/csrc $B/include
for f in $(git -C $ROOT ls-tree --name-only $REV pdp-solver_amd/csrc/); do git -C $ROOT show $REV:$f > $B/csrc/$(basename $f); done
for f in $(git -C $ROOT ls-tree --name-only $REV include/); do git -C $ROOT show $REV:$f > $B/include/$(basename $f); done
sed -i "s#\.\./\.\./include#$B/include#g" $B/csrc/Makefile $B/csrc/*.hpp
make -C $B/csrc -j6 2>&1 | grep -E "error|Error" || true
cp $B/csrc/libpdp_hip.so $ROOT/pdp-solver_amd/csrc/libpdp_hip_prev.so
ls -la $ROOT/pdp-solver_amd/csrc/libpdp_hip_prev.so

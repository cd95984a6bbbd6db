#!/bin/bash
# Per-launch durations of the solver kernels of one bench step, for a library / environment under test (through gpurun):
#   PDP_HIP_LIB=... [ENV=...] bash tools/chunk_trace.sh <tag>
set -u
TAG=${1:-x}
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/chunk_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
rocprofv3 --output-format csv --kernel-trace -d "$OUT/trace" -o trace -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-fast-build > "$OUT/bench.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/trace/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_solve_import' in r['Kernel_Name']]
last = rows[idx[-1]:]
p1 = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in last if 'k_sp_solve_lds<false, false' in r['Kernel_Name']]
p2 = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in last if 'k_sp_solve_lds<false, true' in r['Kernel_Name']]
print('pass 1 us:', ' '.join('%.0f' % x for x in p1), ' sum %.0f' % sum(p1))
print('replay us:', ' '.join('%.0f' % x for x in p2), ' sum %.0f' % sum(p2))
PY

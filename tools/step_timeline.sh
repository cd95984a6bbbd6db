#!/bin/bash
# Timeline of ONE bench step from a rocprofv3 kernel trace (kernels, the runtime's copy / fill kernels, the gaps between them), through gpurun:
#   bash tools/step_timeline.sh <tag>
set -u
TAG=${1:-x}
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/timeline_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
rocprofv3 --output-format csv --kernel-trace -d "$OUT/trace" -o trace -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-fast-build > "$OUT/bench.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
fills = [i for i, r in enumerate(rows) if 'k_bind_fill' in r['Kernel_Name']]
# a step in the middle of the run: from the fills of its bind to the kernel before the next step's
lo, hi = fills[len(fills) // 2], fills[len(fills) // 2 + 1]
t0 = int(rows[lo]['Start_Timestamp']); prev_end = t0
busy = 0.0
for r in rows[lo:hi]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.1f us  +%6.1f gap  %8.1f us  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r['Kernel_Name'][:70]))
    busy += (e - s) / 1e3; prev_end = e
print('step: %.1f us from first to last kernel, %.1f us inside kernels, %.1f us of gaps, %d launches' % ((prev_end - t0) / 1e3, busy, (prev_end - t0) / 1e3 - busy, hi - lo))
PY

#!/bin/bash
# One ad-hoc counter pass over bench.py's headline step: bash tools/pmc_pass.sh <tag> <counters...>  (through gpurun)
# e.g. tools/pmc_pass.sh icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH
set -u
TAG=$1; shift
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
rocprofv3 --output-format csv --kernel-trace --pmc "$@" -d "$OUT/pmc" -o pmc -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-fast-build > "$OUT/bench.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')[:64]
        acc[k][r.get('Counter_Name')] += float(r.get('Counter_Value', 0) or 0); cnt[k].add(r.get('Dispatch_Id'))
for k in sorted(acc, key=lambda x: -sum(acc[x].values()))[:3]:
    n = max(1, len(cnt[k])); print("%-64s dispatches=%d" % (k, n))
    for c, v in sorted(acc[k].items()): print("      %-30s %.5g per dispatch" % (c, v / n))
PY

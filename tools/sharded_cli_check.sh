#!/bin/bash
# Sharded CLI check on a single-GPU box: two ranks share GPU 0 (gloo instead of RCCL), rows and counters must equal the two shards run
# one after the other in single processes.  usage (through gpurun): bash tools/sharded_cli_check.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
OUT=$ROOT/gpurun_out/sharded; rm -rf "$OUT"; mkdir -p "$OUT"
CFG=config/Predict/PDP-p-d-p-walksat-pytorch.yaml
IN=tests/golden/cli_dimacs20.converted.jsonl
PDP_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
    pdp-solver_amd/satyr.py $CFG $IN 30 -z 100 -s 5 --rng philox -v -o $OUT/sharded.jsonl > $OUT/sharded.log 2>&1
python - <<PY
import sys, json
sys.path.insert(0, '$ROOT/pdp-solver_amd')
from pdp import parallel
lines = [l for l in open('$ROOT/$IN').read().split('\n') if l.strip()]
b = parallel.shard_bounds([len(l) for l in lines], 2)
for r, (lo, hi) in enumerate(b):
    open('$OUT/shard%d.json' % r, 'w').write("\n".join(lines[lo:hi]) + "\n")
print('bounds', b)
PY
for r in 0 1; do python pdp-solver_amd/satyr.py $CFG $OUT/shard$r.json 30 -z 100 -s 5 --rng philox -o $OUT/single$r.jsonl > /dev/null 2>&1; done
cat $OUT/single0.jsonl $OUT/single1.jsonl | grep -v '^$' > $OUT/expected.jsonl
grep -v '^$' $OUT/sharded.jsonl > $OUT/got.jsonl
if cmp -s $OUT/expected.jsonl $OUT/got.jsonl; then echo "sharded rows == single-process shards: $(wc -l < $OUT/got.jsonl) rows"; else echo "MISMATCH"; diff $OUT/expected.jsonl $OUT/got.jsonl | head; fi
grep "instances" $OUT/sharded.log | tail -2

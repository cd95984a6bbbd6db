#!/bin/bash
# Sharded CLI check on a single-GPU box: two ranks share GPU 0 (gloo instead of RCCL); rows and counters must equal the UNSHARDED
# single-process run of the same command (whole loader batches are dealt to ranks, random numbers keyed by the global batch index).
# usage (through gpurun): bash tools/sharded_cli_check.sh        (tests/test_sharded_gpu.py is the asserted form of this)
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
OUT=$ROOT/gpurun_out/sharded; rm -rf "$OUT"; mkdir -p "$OUT"
CFG=config/Predict/PDP-p-d-p-sp-pytorch.yaml
IN=tests/golden/cli_dimacs20.converted.jsonl
ARGS="$CFG $IN 30 -z 6 -s 5 -w 40 --rng philox -v"
PDP_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
    pdp-solver_amd/satyr.py $ARGS -o $OUT/sharded.jsonl > $OUT/sharded.log 2>&1
python pdp-solver_amd/satyr.py $ARGS -o $OUT/single.jsonl > $OUT/single.log 2>&1
grep -v '^$' $OUT/single.jsonl > $OUT/expected.jsonl
grep -v '^$' $OUT/sharded.jsonl > $OUT/got.jsonl
if cmp -s $OUT/expected.jsonl $OUT/got.jsonl; then echo "sharded rows == unsharded rows: $(wc -l < $OUT/got.jsonl) rows"; else echo "MISMATCH"; diff $OUT/expected.jsonl $OUT/got.jsonl | head; fi
grep "instances" $OUT/sharded.log | tail -2
grep "instances" $OUT/single.log | tail -1

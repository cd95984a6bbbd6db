#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box: kernel-trace stats, then separate PMC passes
# (counters are never combined with tracing domains other than --kernel-trace).
# usage (through gpurun): bash tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG; rm -rf "$OUT"
mkdir -p "$OUT"
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-fast-build $*"
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/trace" -o trace -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_trace.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d "$OUT/pmc1" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_pmc1.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY -d "$OUT/pmc2" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_pmc2.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_INSTS_SMEM -d "$OUT/pmc5" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_pmc5.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d "$OUT/pmc6" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_pmc6.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_IFETCH SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC -d "$OUT/pmc7" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_pmc7.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc3" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_pmc3.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc4" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_pmc4.log" 2>&1
# the opt-in fast build of the same kernel: kernel stats + the two counter groups that carry its instruction count and mix
export PDP_BUILD=fast
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/fast_trace" -o trace -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_fast_trace.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d "$OUT/fast_pmc1" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_fast_pmc1.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_INSTS_SMEM -d "$OUT/fast_pmc5" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_fast_pmc5.log" 2>&1
unset PDP_BUILD
cd "$ROOT"
python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1
python3 tools/summarize_profile.py "$OUT" fast_ > "$OUT/summary_fast.txt" 2>&1 || true
cat "$OUT/summary.txt"
tail -2 "$OUT/bench_trace.log"

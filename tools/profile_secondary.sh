#!/bin/bash
# rocprofv3 evidence for the kernels BESIDE the headline one: the neural kernels at hidden 128 and 150 (kernel stats +
# the MFMA / VALU / wait counters), the Reinforce instantiation of the persistent solver and the Walk-SAT launch (kernel stats).  Kernel-trace
# stats and each PMC group in runs of their own, never combined with other tracing domains; the program itself behind `--`.
# usage (through gpurun): bash tools/profile_secondary.sh <tag>
set -u
TAG=${1:-r03}
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_sec_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
for H in 128 150; do
  ARGS="--workload neural --hidden $H --iters 3 --steps 2 --warmup 1 --no-cpu-baseline"
  rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/neural$H/trace" -o trace -- python3 "$ROOT/bench.py" $ARGS > "$OUT/neural${H}_trace.log" 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$OUT/neural$H/pmc1" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/neural${H}_pmc1.log" 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU -d "$OUT/neural$H/pmc2" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/neural${H}_pmc2.log" 2>&1
done
# the opt-in fast build (PDP_BUILD=fast: device math on the transcendental unit), hidden 128: kernel stats + the MFMA / VALU counters
ARGS="--workload neural --hidden 128 --iters 3 --steps 2 --warmup 1 --no-cpu-baseline"
export PDP_BUILD=fast
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/neural128_fast/trace" -o trace -- python3 "$ROOT/bench.py" $ARGS > "$OUT/neural128_fast_trace.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$OUT/neural128_fast/pmc1" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/neural128_fast_pmc1.log" 2>&1
unset PDP_BUILD
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/reinforce/trace" -o trace -- python3 "$ROOT/tools/model_time.py" reinforce 100 > "$OUT/reinforce_trace.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/walksat/trace" -o trace -- python3 "$ROOT/tools/walksat_time.py" 200 1000 5000 > "$OUT/walksat_trace.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d "$OUT/walksat/pmc1" -o pmc -- python3 "$ROOT/tools/walksat_time.py" 200 1000 5000 > "$OUT/walksat_pmc1.log" 2>&1
cd "$ROOT"
python3 tools/summarize_secondary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"

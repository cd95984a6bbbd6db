python -m pytest tests/test_hip_solve.py tests/test_hip_fullsize.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2 3; do
for m in 1 0; do
  if [ $m = 1 ]; then export PDP_SOLVE_NO_CARRY_LOGS=1; else unset PDP_SOLVE_NO_CARRY_LOGS; fi
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --no-fast-build 2>>gpurun_out/ab_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('nocarry=$m', d['value'], d['ms_per_step'], c['kernel_ms_per_launch'])"
done; done

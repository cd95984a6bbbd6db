#!/usr/bin/env python3
"""Benchmark of the PDP hot path on MI355X.

Metric (BASELINE.json): PDP message-passing iterations/sec on random 3-SAT n=200 m=840 batch=5000 ('p-d-p' survey
propagation + sequential decimation, T=100, configs[1]).  One "step" = one pass of the hot path over one resident
batch: reset of the solver state, SATProblem.simplify, and the T-iteration propagate/decimate/predict/terminate
loop (reference: src/pdp/nn/solver.py:332-337,355-386) -- the loop runs as ONE persistent kernel launch per chunk of iterations.
value = executed PDP iterations per second, aggregated over all ranks (each rank owns its own batch of 5000
instances: weak scaling, no collective on the data path; one RCCL all-reduce of the solved counters at the end).

Contract: `python bench.py --gpus N --steps K --warmup W`.  With N > 1 and no WORLD_SIZE in the environment this process is only a
launcher: it starts N ranks (one per GPU) through `python -m torch.distributed.run` BEFORE touching the GPU itself, forwards the single
JSON line rank 0 prints, and exits non-zero if any rank fails.  Under torch.distributed.run (what the driver uses for N > 1) it is a
rank and WORLD_SIZE must equal --gpus.  Rank 0 prints ONE JSON line.

Output (benchlib/line.py): the LAST line of stdout is the record -- the contract's top-level keys, `config` of <= 25 scalars, flat `roofline`
and `cpu_baseline`, strict JSON, < 4 KB.  At N = 1 everything else measured outside the timed loop (`cpu_baseline_torch_sparse`, configs[2]'s
kernels with per-kernel rooflines, the configs[3] / configs[4] shards, Walk-SAT, Reinforce, training, solved fractions) is printed BEFORE it as
short `{"detail": name, "data": ...}` lines and written unrounded to gpurun_out/bench_detail.json.  Helpers: tools/benchlib/."""

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))
sys.path.insert(0, os.path.join(REPO, 'tools'))

from benchlib import HBM_PEAK_GBS, N_SIMD, CLOCK_HZ, algorithmic_bytes_per_iteration, grouped       # noqa: E402
from benchlib.cpu_baselines import cpu_baseline_all_cores, cpu_baseline_torch_sparse                  # noqa: E402
from benchlib.neural import bench_neural                                                              # noqa: E402
from benchlib.line import emit                                                                        # noqa: E402
from benchlib.secondary import (fast_build_measurement, secondary_measurements, config_shard_measurements, big_instance_measurements,   # noqa: E402
                                solved_fractions, driver_summary)


# =====================================================================================================================
# launcher: `python bench.py --gpus N` with N > 1 outside torch.distributed.run
# =====================================================================================================================
def launch_ranks(n, argv):
    """Start n ranks of this script (one per GPU) and forward rank 0's JSON line.  Nothing here touches the GPU: a process that has
    initialised HIP must not fork / exec ranks, so the launch happens before any device call."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, env=env)
    lines = [l for l in proc.stdout.split('\n') if l.startswith('{') and '"metric"' in l]
    if proc.returncode != 0 or not lines:
        sys.stderr.write(proc.stdout[-4000:] + '\n' + proc.stderr[-8000:] + '\n')
        sys.stderr.write("bench.py: the %d-rank run failed (exit code %d)\n" % (n, proc.returncode))
        return proc.returncode or 1
    line = json.loads(lines[-1])
    if line.get('rccl_ranks') != n or line.get('n_gpus') != n:
        sys.stderr.write("bench.py: asked for %d ranks, the line reports %r\n" % (n, (line.get('n_gpus'), line.get('rccl_ranks'))))
        return 1
    print(lines[-1])
    return 0


def init_ranks(args):
    """(world, rank, local_rank, backend) of this process; initialises torch.distributed when grouped().  backend 'nccl' is RCCL (one GPU
    per rank); PDP_DIST_BACKEND=gloo lets several ranks share one GPU (checks of the N > 1 code on a single-GPU box / on CPU)."""
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    backend = os.environ.get('PDP_DIST_BACKEND', 'nccl')
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d ranks were started" % (args.gpus, world))
    if grouped():
        import torch.distributed as dist
        if args.selftest_collective:
            dist.init_process_group(backend if backend != 'nccl' or torch.cuda.is_available() else 'gloo')
        elif backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            local_rank = local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: process group of %d ranks for --gpus %d" % (dist.get_world_size(), args.gpus))
    return world, rank, local_rank, backend


def selftest_collective(args):
    """The N-rank plumbing without any GPU work (tests/test_parallel_gloo.py runs it with gloo on CPU): rendezvous, barrier, MAX of the
    per-rank times, SUM of the per-rank counters, rank 0's single line, non-zero exit when a rank fails."""
    import torch
    import torch.distributed as dist
    world, rank, _, _ = init_ranks(args)
    if os.environ.get('PDP_BENCH_FAIL_RANK') == str(rank):
        raise SystemExit("bench.py selftest: rank %d fails on request" % rank)
    elapsed = 0.010 * (rank + 1)
    stats = torch.tensor([1000.0, float(rank + 1), 100.0 * args.steps], dtype=torch.float64)
    ranks = 1
    if grouped():
        dist.barrier()
        tmax = torch.tensor([elapsed], dtype=torch.float64); dist.all_reduce(tmax, op=dist.ReduceOp.MAX); elapsed = float(tmax.item())
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        ranks = dist.get_world_size()
    if rank == 0:
        print(json.dumps({'metric': 'pdp_iterations_per_sec', 'selftest': True, 'value': float(stats[2].item()) / elapsed, 'n_gpus': world,
                          'rccl_ranks': ranks, 'steps': args.steps, 'warmup': args.warmup, 'instances': float(stats[0].item()),
                          'rank_sum': float(stats[1].item()), 'max_elapsed_s': elapsed}))
    if grouped():
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=5000)
    ap.add_argument('--n', type=int, default=200)
    ap.add_argument('--iters', type=int, default=100)
    ap.add_argument('--tolerance', type=float, default=0.02)
    ap.add_argument('--t_max', type=float, default=100)
    ap.add_argument('--walksat', type=int, default=100, help='Walk-SAT steps of the (untimed) solved-fraction pass')
    ap.add_argument('--workload', choices=['sp', 'neural'], default='sp',
                    help="sp: configs[1] (headline metric); neural: configs[2] 'np-nd-np' hidden_dim=128 on the same graph (fp32 MFMA)")
    ap.add_argument('--hidden', type=int, default=128)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fast-build', action='store_true', help='skip the rerun of the headline loop on the opt-in fast build (config.fast_build)')
    ap.add_argument('--no-secondary', action='store_true', help='skip config.secondary (neural / Walk-SAT / Reinforce) and the long solved-fraction runs')
    ap.add_argument('--isolated', action='store_true', help="sp workload: every instance solved on its own (no batch-wide couplings of the "
                    "reference, hence no NaN-poison replay); the default is the reference's strict semantics")
    ap.add_argument('--seed-rank', type=int, default=None, help='generate the batch another rank would get (checks of the sharded run on one GPU)')
    ap.add_argument('--cpu-sample-batch', type=int, default=1000, help='CPU baseline without --cpu-full-batch: instances of the sample')
    ap.add_argument('--cpu-cores', type=int, default=0, help='cap the CPU baselines at this many cores (0: all)')
    ap.add_argument('--cpu-partial-batch', dest='cpu_full_batch', action='store_false', help='CPU baselines on --cpu-sample-batch instances only')
    ap.add_argument('--cpu-torch-budget-s', type=float, default=25.0, help='torch sparse-mm baseline at the full batch: run the steady-state '
                    'iterations only if set-up + first iteration took less than this many seconds')
    ap.add_argument('--secondary-neural-iters', type=int, default=3)
    ap.add_argument('--secondary-walksat-steps', type=int, default=1000)
    ap.add_argument('--config3-batch', type=int, default=5000, help='instances of the configs[3] shard measurement (n = 400)')
    ap.add_argument('--train-batch', type=int, default=400, help='instances of the training-step measurement (n = --n)')
    ap.add_argument('--config4-instances', type=int, default=600, help='instances (before the 4 replicas) of the configs[4] shard measurement')
    ap.add_argument('--selftest-collective', action='store_true', help='the N-rank plumbing only (no GPU work); used by the gloo test')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.selftest_collective:
        return selftest_collective(args)

    world_env, rank_env = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    m = int(round(4.2 * args.n))
    from pdp.factorgraph import dataset
    items = None
    cpu, cpu_ts = None, None
    if args.workload == 'sp':
        items = dataset.random_ksat_items(args.batch, args.n, 3, m=m, seed=1000003 * (rank_env if args.seed_rank is None else args.seed_rank))
        if world_env == 1 and not args.no_cpu_baseline:
            # before anything initialises the GPU in this process: the oracle workers are forked
            cpu = cpu_baseline_all_cores(args, items)
            cpu_ts = cpu_baseline_torch_sparse(args, items)

    import torch
    from pdp import native
    world, rank, local_rank, backend = init_ranks(args)
    native.require_gpu()
    dev = torch.device('cuda', local_rank if world > 1 else 0)
    args.coll_dev = dev if backend == 'nccl' else torch.device('cpu')
    torch.cuda.set_device(dev)
    if args.workload == 'neural':
        return bench_neural(args, dev, rank, world)

    # ---- synthetic batch, resident in HBM before the timed region ---------------------------------------------
    host_batch = dataset.collate_segment(items)
    torch.cuda.synchronize(); t_setup = time.perf_counter()
    b = dataset.to_torch(host_batch, dev)                      # PCIe upload of the loader tensors
    prob = native.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'], batch_size=args.batch)
    torch.cuda.synchronize(); setup_ms = 1e3 * (time.perf_counter() - t_setup)   # reported, never part of `value`
    E, V, F, B = prob.E, prob.V, prob.F, prob.B
    # the inputs of every step are resident before the timed region starts: one initial state (get_init_state(randomized=False),
    # pdp_propagate.py:233-235), active mask and decimator per step -- 0.25 GB each
    n_states = args.warmup + args.steps + 3          # + the three steps with events around the whole call
    states = []
    for _ in range(n_states):
        q_ = torch.full((E, 3), 1.0, device=dev); q_.div_(3.0)
        fs_ = torch.zeros(E, 2, device=dev); fs_[:, 0] = 0.5
        states.append((q_, fs_, torch.ones(B, dtype=torch.uint8, device=dev), native.Decimator(prob)))
    q, fs, am, dec = states[0]
    L = native.lib()
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    call_ms, iters_done, paths, launches = [], [], [], []

    step_no = [0]
    # (the five state arrays and the stream do not change from step to step: their ctypes forms are made once)
    bind_args = (native.ptr(prob.active_variables), native.ptr(prob.active_functions), native.ptr(prob.solution), native.ptr(prob.is_sat),
                 native.ptr(prob.edge_mask), native._stream())

    def step(record, call_events=False):
        # a fresh SATProblem (solver.py:49-54: the library resets the problem's state arrays) + simplify(), then the solver on this step's
        # initial state.  The timed steps carry the library's events around the chunk launches only (an event record between two dependent
        # launches costs 3-6 us of stream time, tools/micro/event_gap.hip); the events around the whole call go on extra steps behind the timed region
        q, fs, am, dec = states[step_no[0] % n_states]
        step_no[0] += 1
        native.check(L.pdp_problem_bind_state(prob._h, *bind_args))
        prob.simplify()
        if call_events: ev0.record()
        try:
            it, lds = prob.sp_solve(q, fs, am, dec, args.iters, args.tolerance, args.t_max, time_kernels=True, isolate_instances=args.isolated,
                                    inputs_disposable=True)      # like the solver class: this step's q / fs are copies of the initial state
            path = 'persistent-lds' if lds else 'persistent-hbm'
        except native.SpeculationFailed:
            raise SystemExit("bench: speculation failed on the benchmark batch (unexpected)")
        if call_events:
            ev1.record(); torch.cuda.synchronize()
            call_ms.append(ev0.elapsed_time(ev1))
        if record:
            torch.cuda.synchronize()
            iters_done.append(it); paths.append(path); launches.append(dict(prob.last_solve_stats))

    def barrier():
        if grouped():
            import torch.distributed as dist
            dist.barrier()

    for _ in range(args.warmup):
        step(False)
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize(); barrier()
    elapsed = time.perf_counter() - t0
    total_iters = float(sum(iters_done))
    for _ in range(3):                                  # untimed: HIP events around the whole pdp_sp_solve call (config.solve_call_ms)
        step(False, call_events=True)
    del states[:]
    torch.cuda.empty_cache()

    # ---- untimed: finish the forward pass once (random fill + Walk-SAT) for the solved fraction ------------------
    prob.random_fill(seed=12345 + rank)
    out, ws_steps = prob.local_search(prob.solution.clone(), args.walksat, 0.5, seed=999 + rank)
    pred = prob.update_solution(out.reshape(-1).contiguous())
    solved, unsat = prob.cnf_eval(pred.reshape(-1).contiguous())
    stats = torch.tensor([float(B), float(solved.sum().item()), float(unsat.sum().item()), elapsed, total_iters],
                         dtype=torch.float64, device=args.coll_dev)
    ranks = 1
    if grouped():
        import torch.distributed as dist
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=args.coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)       # the only collective of the path: a 40-byte sum over xGMI
        ranks = dist.get_world_size()
    n_inst, n_solved, n_unsat, _, iters_all = [float(x) for x in stats.tolist()]

    if rank == 0:
        value = iters_all / elapsed
        kms = float(np.mean(call_ms))                   # HIP-event time of one pdp_sp_solve call (all its launches), three steps behind the timed ones
        it_mean = float(np.mean(iters_done))
        n_launch = float(np.mean([l['launches'] for l in launches]))
        n_replay = float(np.mean([l['replays'] for l in launches]))
        # dominant kernel k_sp_solve_lds<false, false, false> (one launch per chunk of iterations): HIP events recorded by the library
        # on the launch stream around every launch; algorithmic bytes = bytes/iteration x iterations per launch
        launch_ms = float(np.mean([l['solve_kernel_ms'] for l in launches])) / n_launch
        if launch_ms <= 0.0:                            # HBM-resident fallback kernel (instances too large for the LDS): no per-launch events,
            launch_ms = kms / max(n_launch, 1.0)        # the events around the whole call divided by its launches
        replay_ms = float(np.mean([l['replay_kernel_ms'] for l in launches]))
        bytes_launch = algorithmic_bytes_per_iteration(E, V, F) * it_mean / n_launch
        achieved = bytes_launch / (launch_ms * 1e-3) / 1e9
        # measured HBM traffic and VALU instruction count per launch: NOT measured in this run -- read from the committed rocprofv3 PMC
        # summary of the same kernel and labelled with their source
        traffic, valu, src = None, None, None
        pmcs = sorted(f for f in os.listdir(os.path.join(REPO, 'profiles')) if f.endswith('_pmc_traffic.json')) if os.path.isdir(os.path.join(REPO, 'profiles')) else []
        if pmcs and args.batch == 5000 and args.n == 200:
            try:
                src = 'profiles/' + pmcs[-1]
                pj = json.load(open(os.path.join(REPO, src)))
                # (per launch of the profiled build; a call's work is the same whatever the chunk length, so both scale with the launches per call)
                scale = float(pj.get('launches_per_call', 9.0)) / max(n_launch, 1.0)
                traffic = pj.get('k_sp_solve_lds_bytes_per_launch')
                if traffic: traffic = float(traffic) * scale
                if pj.get('SQ_INSTS_VALU_per_launch'):
                    # what binds the LDS-resident kernel: wave-level VALU instructions against the issue slots of 1024 SIMDs over the measured
                    # launch time.  A wave64 fp32 instruction occupies its SIMD for 2 cycles when the same wave has an independent instruction
                    # next and ~4 in a dependent chain (tools/micro/pk_rate.hip: 2.2 / 4.3 measured; MI355X_MICROARCH.md: 2 cycles) -- both given
                    insts = float(pj['SQ_INSTS_VALU_per_launch']) * scale
                    cyc = N_SIMD * CLOCK_HZ * launch_ms * 1e-3
                    valu = {'insts_per_launch': insts, 'cycles_per_inst_per_simd': cyc / insts, 'issue_frac_at_2_cycles': insts * 2.0 / cyc,
                            'issue_frac_at_4_cycles_dependent_chain': insts * 4.0 / cyc, 'source': src,
                            'note': 'SQ_INSTS_VALU of a rocprofv3 PMC pass of the same kernel (committed summary, not this run) / (1024 SIMDs x 2.4 GHz x '
                                    'this run\'s launch time)'}
            except Exception:
                traffic, valu = None, None
        solve_kernel_name = native.kernel_name('sp_solve')
        config = {'workload': "configs[1]: 'p-d-p' survey propagation, random 3-SAT n=%d m=%d batch=%d T=%d per GPU" % (args.n, m, args.batch, args.iters),
                  'E': E, 'V': V, 'F': F, 'iterations_per_step': it_mean, 'path': paths[0] if paths else None,
                  'instance_iterations_per_sec': value * args.batch, 'edge_updates_per_sec': value * 2 * E,
                  'solve_call_ms': kms, 'kernel_launches_per_call': n_launch, 'kernel_ms_per_launch': launch_ms,
                  'poison_replay_launches_per_call': n_replay, 'poison_replay_ms_per_call': replay_ms,
                  'algorithmic_bytes_per_launch': bytes_launch, 'setup_ms_upload_and_layout': setup_ms, 'solved_fraction': n_solved / n_inst, 'unsat_clauses_total': n_unsat,
                  'walksat_steps': ws_steps, 'tolerance': args.tolerance, 't_max': args.t_max, 'parallelism': 'instances sharded, dp%d' % world,
                  'semantics': 'isolated instances' if args.isolated else "reference (batch-wide couplings reproduced)"}
        if world == 1 and not args.no_secondary:
            config['solved'] = solved_fractions(args, dev, b, native, rank)
            config['secondary'] = secondary_measurements(args, dev, b, prob, native, items)
            config['secondary'].update(big_instance_measurements(args, dev, items, value, native))
            del prob, b
            torch.cuda.empty_cache()
            config['secondary'].update(config_shard_measurements(args, dev, native))
        if world == 1 and not args.no_fast_build:
            try:
                prob = b = None
                torch.cuda.empty_cache()
                config['fast_build'] = fast_build_measurement(args, dev, native, host_batch, items, n_solved / n_inst)
            except Exception as ex:
                config['fast_build'] = dict(error=repr(ex))
        summary = driver_summary(config)                # flat scalars of every BASELINE config (detail line "summary"; ten of them in the final line)
        full = {
            'metric': 'pdp_iterations_per_sec', 'value': value,
            'unit': 'iterations/s (each iteration sweeps a batch of %d instances)' % args.batch,
            'n_gpus': world, 'rccl_ranks': ranks, 'collective_backend': (backend if grouped() else None), 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': config,
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'traffic_source': src, 'valu_issue': valu, 'kernel': solve_kernel_name,
                         'note': 'achieved = streaming-model algorithmic bytes (41E+36V+8F per iteration) x iterations per launch / '
                                 'average launch duration (HIP events on the launch stream, this run); the instance state is LDS-resident, '
                                 'so the kernel is bound by VALU issue, not by HBM (DESIGN.md section 4); traffic / valu_issue come from the '
                                 'committed PMC summary named in their source fields, not from this run'},
            'cpu_baseline': cpu, 'cpu_baseline_torch_sparse': cpu_ts,
            'summary': summary,
        }
        emit(full, side_file=(world == 1))              # short {"detail": ...} lines, then the compact record as the LAST line of stdout
    if grouped():
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

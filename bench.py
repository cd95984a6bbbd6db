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

At N = 1 the line also carries, outside the headline's timed loop: `cpu_baseline` (the C oracle over all host cores at the full batch;
`cpu_baseline_torch_sparse`: the PyTorch-CPU restatement of the reference's sparse-mm formulation) and `config.secondary`
(configs[2]'s neural kernels with per-kernel rooflines, Walk-SAT, the Reinforce solver)."""

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA = the fp32 vector peak
N_SIMD, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs, 2.4 GHz


def algorithmic_bytes_per_iteration(E, V, F):
    "SURVEY.md section 8(d): streaming model of one SP iteration, 41 B/edge + 36 B/variable + 8 B/clause"
    return 41 * E + 36 * V + 8 * F


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


def grouped():
    """Does this process join a torch.distributed group?  Always with several ranks; with ONE rank only on request (PDP_DIST_FORCE=1 under
    torch.distributed.run --nproc-per-node 1): the barrier and the two all-reduces then go through RCCL on a one-GPU box exactly as they
    do on eight, and the line says rccl_ranks = 1 with the backend that ran."""
    return int(os.environ.get('WORLD_SIZE', '1')) > 1 or (os.environ.get('PDP_DIST_FORCE') == '1' and 'RANK' in os.environ)


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


# =====================================================================================================================
# CPU baselines (rank 0, N = 1; run BEFORE the GPU is touched: the oracle workers are forked)
# =====================================================================================================================
_CPU_ITEMS = None


def _oracle_worker(job):
    lo, hi, iters, tol, t_max = job
    from oracle import binding
    from pdp.factorgraph import dataset
    b = dataset.collate_segment(_CPU_ITEMS[lo:hi])
    p = binding.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
    res = p.forward('p-d-p', iters, local_search_iterations=0, tolerance=tol, t_max=t_max, seed=1)
    return (hi - lo) * res['iterations_run']


def effective_cores():
    """host cores this process may really use: min(os.cpu_count(), the scheduler affinity mask, the cgroup CPU quota).  On the GPU boxes of
    this pool os.cpu_count() reports every hardware thread of the host (256) while the container's quota is far smaller; worker pools and
    torch thread counts sized by cpu_count() then oversubscribe and run many times slower than one thread per usable core."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]                      # cgroup v2
        if q != 'max':
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read()); per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)


def cpu_model_name():
    try:
        for l in open('/proc/cpuinfo'):
            if l.startswith('model name'):
                return l.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline_all_cores(args, items):
    """The C oracle (single-threaded restatement of the reference's algorithm, oracle/pdp_oracle.c) on EVERY host core: the full batch of
    the headline workload cut into one contiguous sub-batch per worker process, every sub-batch run for the full T iterations.  (Each
    sub-batch is a batch of its own for the reference's batch-wide couplings; this is a throughput baseline.)"""
    import multiprocessing as mp
    global _CPU_ITEMS
    sys.path.insert(0, REPO)
    from oracle import binding
    binding.build()
    cores = effective_cores()
    if args.cpu_cores:
        cores = min(cores, args.cpu_cores)
    B = len(items) if args.cpu_full_batch else min(len(items), args.cpu_sample_batch)
    _CPU_ITEMS = items[:B]
    workers = min(cores, B)
    bounds = [(B * w) // workers for w in range(workers + 1)]
    jobs = [(bounds[w], bounds[w + 1], args.iters, args.tolerance, args.t_max) for w in range(workers)]
    ctx = mp.get_context('fork')
    with ctx.Pool(workers) as pool:
        pool.map(_oracle_worker, [(0, 1, 1, args.tolerance, args.t_max)] * workers)       # start the workers, load the library
        t0 = time.perf_counter()
        done = pool.map(_oracle_worker, jobs, chunksize=1)
        dt = time.perf_counter() - t0
    _CPU_ITEMS = None
    inst_iters = float(sum(done)) / dt
    return dict(value=inst_iters / args.batch, unit='iterations/s (batch of %d instances)' % args.batch, cores=workers, kind='port',
                cpu_model=cpu_model_name(), host_hardware_threads=os.cpu_count(),
                sample='the oracle on %d worker processes (one per usable host core: min of cpu_count %d, affinity, cgroup quota), %d instances x '
                       '%d iterations of the headline batch in %.2f s (%.0f instance-iterations/s)%s'
                       % (workers, os.cpu_count() or 1, B, args.iters, dt, inst_iters, '' if B == args.batch else ', scaled linearly to the batch'))


def cpu_baseline_torch_sparse(args, items):
    """The reference's own formulation on the CPU: sparse COO masks + torch.mm + the dense [V x B] matrices of sparse_max / sparse_argmax
    (oracle/torch_sparse_port.py, an own restatement of the op sequence; the reference itself cannot travel to this box), with
    torch.set_num_threads(all cores) as src/pdp/factorgraph/base.py:43-50 does.  B = 500 for 3 iterations, then the full batch: its cost is
    quadratic in B (20 GB dense matrix per reduction at B = 5000), so the full batch runs 2 iterations and the second one -- the first with
    a convergence test -- is the per-iteration figure."""
    import torch
    sys.path.insert(0, REPO)
    from oracle import torch_sparse_port as port
    from pdp.factorgraph import dataset
    cores = effective_cores()
    if args.cpu_cores:
        cores = min(cores, args.cpu_cores)
    torch.set_num_threads(cores)
    out = dict(unit='iterations/s (batch of B instances)', cores=cores, kind='port', cpu_model=cpu_model_name(), runs=[])
    def mem_available_gb():
        try:
            for l in open('/proc/meminfo'):
                if l.startswith('MemAvailable'):
                    return float(l.split()[1]) / 1e6
        except OSError:
            pass
        return 0.0

    for B, T in ((500, 3), (args.batch, 2)):
        if B > len(items) or (B > 500 and not args.cpu_full_batch):
            continue
        need_gb = 3.0 * 4e-9 * B * (B * args.n)                     # three live dense [V x B] fp32 matrices at the worst point
        if B > 500 and mem_available_gb() < need_gb + 16.0:
            out['skipped'] = 'B=%d needs ~%.0f GB of host memory for the dense [V x B] matrices (%.0f GB available)' % (B, need_gb, mem_available_gb())
            continue
        b = dataset.collate_segment(items[:B])
        t0 = time.perf_counter()
        P = port.SparseBatch(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
        with torch.no_grad():
            P.simplify()
            setup = time.perf_counter() - t0
            times = []
            # the full batch stops after the first iteration when set-up + that iteration exceed the budget of the default run (the
            # steady-state iterations add the convergence test: two more dense [V x B] reductions each)
            port.forward_loop(P, T, tolerance=args.tolerance, t_max=args.t_max, simplify=False, times=times,
                              max_seconds=None if B <= 500 else max(0.0, args.cpu_torch_budget_s - setup))
        steady = times[1:] if len(times) > 1 else times
        per_iter = float(np.mean(steady))
        out['runs'].append(dict(B=B, iterations=len(times), seconds_per_iteration=per_iter, first_iteration_s=times[0], setup_s=setup,
                                iterations_per_s=1.0 / per_iter, instance_iterations_per_s=B / per_iter))
        del P
    if out['runs']:
        last = out['runs'][-1]
        out['value'] = last['iterations_per_s'] * last['B'] / float(args.batch) if last['B'] != args.batch else last['iterations_per_s']
        out['sample'] = 'torch sparse-mm restatement, %d threads: ' % cores + '; '.join(
            'B=%d: %.2f s per iteration (%d iterations run)' % (r['B'], r['seconds_per_iteration'], r['iterations']) for r in out['runs'])
    return out


def cpu_baseline_neural(args):
    """The oracle's operators of one np-nd-np iteration (2 edge aggregators, 2 GRU cells, predictor aggregator + head), single thread,
    on a few instances of the same family with random weights of the same shapes; scaled linearly to the batch."""
    sys.path.insert(0, REPO)
    from oracle import binding
    from pdp.factorgraph import dataset
    binding.build()
    H = args.hidden
    bs = max(1, min(400, args.batch, int(400 * 200 / max(1, args.n))))       # ~1 M edges however large the instances are (400 instances at n = 200)
    b = dataset.collate_segment(dataset.random_ksat_items(bs, args.n, 3, m=int(round(4.2 * args.n)), seed=777))
    gm = np.asarray(b['graph_map']); ev, ec = gm[0].astype(np.int32), gm[1].astype(np.int32)
    es = np.asarray(b['edge_feature'], dtype=np.float32).reshape(-1)
    E, V, F = ev.size, int(np.asarray(b['batch_variable_map']).size), int(np.asarray(b['batch_function_map']).size)
    rng = np.random.RandomState(1)
    r = lambda *sh: (rng.randn(*sh) * 0.2).astype(np.float32)
    agg = lambda fd: dict(W1m=r(100, H + 1), b1m=r(100), W2m=r(50, 100), W1a=r(100, 50 + fd), b1a=r(100), W2a=r(H, 100))
    wv, wf, wp = agg(1), agg(1), agg(0)
    gv = dict(W_ih=r(3 * H, H + 1), W_hh=r(3 * H, H), b_ih=r(3 * H), b_hh=r(3 * H)); gf = dict(gv)
    head = (r(50, H), r(50), r(1, 50))
    dv, df, pv, pf = r(E, H), r(E, H), r(E, H), r(E, H)
    t0 = time.perf_counter()
    pf2 = binding.aggregator(ev, V, dv, es, None, False, wv); pv2 = binding.aggregator(ec, F, df, es, None, False, wf)
    dv2 = binding.gru(pv2, es, dv, **gv); df2 = binding.gru(pf2, es, df, **gf)
    binding.perceptron(binding.aggregator(ev, V, dv2, es, None, True, wp), *head)
    dt = time.perf_counter() - t0
    return dict(value=bs / dt / args.batch, unit='iterations/s (batch of %d instances)' % args.batch, cores=1, kind='port',
                sample='one np-nd-np iteration (2 aggregators, 2 GRU cells, predictor) of %d instances of the same n=%d family, hidden %d, '
                       'in %.1f s, scaled linearly to the batch' % (bs, args.n, H, dt))


# =====================================================================================================================
# configs[2]: the neural workload (standalone with --workload neural, and as a short secondary measurement of the default run)
# =====================================================================================================================
# MACs per edge / per variable of the kernels of one neural iteration at hidden H, inner widths 100 / 50 / 100
def neural_flops(H, model_type='np-nd-np'):
    "flop per launch unit: per EDGE for agg_pre / agg_post / gru (one cell), per VARIABLE for predict_head"
    gru_in = 2.0 * (3 * H * (H + 1) + 3 * H * H)                      # np-nd-np: [state, sign] -> 129 inputs at H = 128
    if model_type == 'p-nd-np':                                       # surveys + sign / [eta, force] + sign: 4- and 3-wide inputs, mean of the two cells
        gru_in = 2.0 * (3 * H * 3.5 + 3 * H * H)
    return dict(agg_pre=2.0 * ((H + 1) * 100 + 100 * 50),            # W1_m, W2_m
                agg_post=2.0 * (51 * 100 + 100 * H),                 # W1_a, W2_a
                gru=gru_in,                                          # W_ih, W_hh of ONE cell
                predict_head=2.0 * (50 * 100 + 100 * H + H * 50 + 50))    # predictor's W1_a, W2_a + perceptron head


def neural_flop_per_iteration(model_type, H, E, V):
    """algorithmic flop of one iteration (573 752 E + 48 500 V for np-nd-np at H = 128).  np-nd-np: two edge aggregators, two GRU
    cells, the predictor's pre-transform + per-variable layers.  p-nd-np: the propagator is the SP sweep (no matrix work) behind three
    H-long dot products per edge (the adaptors), GRU cells with 4- and 3-wide inputs, the same predictor."""
    fl = neural_flops(H, model_type)
    if model_type == 'np-nd-np':
        per_edge = 2 * (fl['agg_pre'] + fl['agg_post']) + 2 * fl['gru'] + fl['agg_pre']
    else:
        per_edge = 2.0 * 3 * H + 2 * fl['gru'] + fl['agg_pre']
    return per_edge * E + fl['predict_head'] * V


def make_neural_model(args, T, model_type='np-nd-np', hidden=None):
    import logging
    import torch
    from pdp.trainer import SatFactorGraphTrainer
    cfg = dict(model_type=model_type, model_name='bench-' + model_type, verbose=False, local_search_iteration=0, epsilon=0.5, rng='philox',
               random_seed=1, hidden_dim=hidden or args.hidden, edge_feature_dim=1, meta_feature_dim=0, prediction_dim=1, mem_hidden_dim=100,
               agg_hidden_dim=100, mem_agg_hidden_dim=50, classifier_dim=50, test_batch_limit=1 << 62, batch_size=args.batch,
               test_recurrence_num=T, tolerance=args.tolerance, t_max=args.t_max)
    torch.manual_seed(1234)
    tr = SatFactorGraphTrainer(cfg, use_cuda=True, logger=logging.getLogger('bench'))
    return tr, tr._model_list[0]


def neural_step(tr, model, b, T, replication=1):
    import torch
    from pdp.nn.solver import OwnedState
    gm, bvm, bfm, ef = b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature']
    with torch.no_grad():
        # exactly what FactorGraphTrainerBase._predict_batch does: the initial state is handed over, not kept
        model.forward(init_state=OwnedState(model.get_init_state(gm, bvm, bfm, ef, None, randomized=False, batch_replication=replication)),
                      graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                      is_training=False, iteration_num=T, check_termination=tr._check_recurrence_termination, batch_replication=replication)
    return model.last_run['iterations']


def neural_kernel_rooflines(native, timing, E, V, H, model_type='np-nd-np'):
    """per-kernel lines from the library's HIP events: ms per launch, algorithmic flop per launch, TFLOP/s, fraction of the fp32 MFMA peak.
    E / V: edges / variables one launch covers (summed over launches when segments differ: pass the launch-weighted means).  The kernel
    names are what the library reports it launched last (pdp_kernel_name), not literals."""
    fl = neural_flops(H, model_type)
    out = {}
    for key in ('agg_pre', 'agg_post', 'gru', 'predict_head', 'row_sum', 'sp_adaptors', 'sp_sweep'):
        ms, n = timing[key]
        if n == 0:
            continue
        per = ms / n
        row = dict(kernel=native.kernel_name(key), launches=n, ms_per_launch=per)
        if key == 'sp_adaptors':                          # HBM-bound: the two [E, H] decimator states read once
            gbs = 2.0 * E * H * 4 / (per * 1e-3) / 1e9
            row.update(bytes_per_launch=2.0 * E * H * 4, gb_per_s=gbs, frac_of_hbm_peak=gbs / HBM_PEAK_GBS)
        if key in fl:
            flop = fl[key] * (V if key == 'predict_head' else E)
            tf = flop / (per * 1e-3) / 1e12
            row.update(flop_per_launch=flop, tflops=tf, frac_of_mfma_f32_peak=tf / MFMA_F32_PEAK_TFLOPS)
        out[key] = row
    if 'agg_pre' in out and 'agg_post' in out:
        # one MessageAggregator call of the propagator = pre + row sum + post (the pre launches also serve the predictor: per launch figures)
        ms = out['agg_pre']['ms_per_launch'] + out['agg_post']['ms_per_launch'] + out.get('row_sum', {}).get('ms_per_launch', 0.0)
        tf = (fl['agg_pre'] + fl['agg_post']) * E / (ms * 1e-3) / 1e12
        out['aggregator_call'] = dict(ms=ms, tflops=tf, frac_of_mfma_f32_peak=tf / MFMA_F32_PEAK_TFLOPS)
    return out


def neural_shard(args, dev, native, items, model_type, hidden, T, replication=1, limit=None, walksat_steps=0, workload=''):
    """One rank's share of a neural BASELINE config on this GPU, outside the headline's timed loop: the loader's dynamic segments (dataset.divide
    with the reference's edge x hidden limit), T sweeps of the model per segment through the Python API (warm-up pass first), then the
    Walk-SAT pass on the last segment's problem.  Returns the numbers every fraction is computed from."""
    import torch
    from pdp.factorgraph import dataset
    edges = [it[2].shape[1] for it in items]
    segs = dataset.divide(edges, (limit or (1 << 62)) // replication, hidden)
    tr, model = make_neural_model(args, T, model_type, hidden)
    batches = [dataset.to_torch(dataset.collate_segment([items[j] for j in seg]), dev) for seg in segs]
    E_seg = [int(b['graph_map'].size(1)) * replication for b in batches]
    V_seg = [int(b['batch_variable_map'].numel()) * replication for b in batches]
    for b in batches:                                                  # warm-up: the same pass once (native workspaces, torch's caching allocator)
        neural_step(tr, model, b, T, replication)
    # best of two timed passes: at 25 M edges every [E, 128] state is 12.9 GB and torch's caching allocator may still release and re-acquire
    # blocks in the first pass after the warm-up (a forward then takes 2-3 x its steady-state time; tools/neural_forward_phases.py)
    dt, its, timing = None, None, None
    torch.cuda.reset_peak_memory_stats()
    for _ in range(2):
        torch.cuda.synchronize()
        native.kernel_timing(True)
        t0 = time.perf_counter()
        its_ = [neural_step(tr, model, b, T, replication) for b in batches]
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        timing_ = native.kernel_timing_read(); native.kernel_timing(False)
        if dt is None or dt_ < dt:
            dt, its, timing = dt_, its_, timing_
    flop = sum(neural_flop_per_iteration(model_type, hidden, e, v) * it for e, v, it in zip(E_seg, V_seg, its))
    tf = flop / dt / 1e12
    n_seg = float(len(segs))
    out = dict(workload=workload, model_type=model_type, hidden=hidden, instances=len(items), batch_replication=replication,
               segments=[len(sg) for sg in segs], edges_per_segment_with_replicas=E_seg, iterations_per_segment=its, seconds=dt,
               segment_iterations_per_sec=sum(its) / dt, ms_per_iteration_mean=1e3 * dt / max(1, sum(its)), flop_total=flop,
               flop_per_iteration_mean=flop / max(1, sum(its)), path=model.last_run['path'],
               # device memory of the timed passes: torch's allocator (the [E, H] states; the library's own workspaces are not in it)
               max_memory_reserved_gb=torch.cuda.max_memory_reserved() / 1e9, max_memory_allocated_gb=torch.cuda.max_memory_allocated() / 1e9,
               state_tensor_gb=max(E_seg) * hidden * 4 / 1e9,
               roofline=dict(bound='mfma', achieved=tf, peak=MFMA_F32_PEAK_TFLOPS, unit='TFLOP/s', frac=tf / MFMA_F32_PEAK_TFLOPS,
                             note='algorithmic flop of the executed sweeps (neural_flop_per_iteration per segment) / wall time of the forwards, '
                                  'set-up of each SATProblem included'),
               kernels=neural_kernel_rooflines(native, timing, sum(E_seg) / n_seg, sum(V_seg) / n_seg, hidden, model_type))
    if walksat_steps > 0:
        prob = model._last_problem._native
        prob.random_fill(seed=4321)
        start = prob.solution.clone()
        prob.local_search(start, 2, 0.5, seed=5)
        torch.cuda.synchronize()
        native.kernel_timing(True)
        t0 = time.perf_counter()
        res, steps = prob.local_search(start, walksat_steps, 0.5, seed=999)
        torch.cuda.synchronize()
        dtw = time.perf_counter() - t0
        kms, kn = native.kernel_timing_read()['walksat']; native.kernel_timing(False)
        out['walksat'] = dict(steps=steps, instances_with_replicas=prob.B, call_seconds=dtw, kernel=native.kernel_name('walksat'), kernel_ms=kms,
                              kernel_launches=kn, flips_per_sec=steps * prob.B / dtw, us_per_step=1e6 * dtw / max(1, steps))
    del tr, model, batches
    torch.cuda.empty_cache()
    return out


def train_measurement(args, dev, native):
    """one optimizer step (`_train_batch`, base.py:149-182) per model type that trains, on a ~1 M-edge batch at hidden 128 --
    3 outer recurrences, random initial states and dropout 0.2 from the device generator, clipped Adam step.  flop = 3 x the forward's algorithmic flop (the adjoint
    of every dense layer is two products of the forward's size) x recurrences; the fraction is against the fp32 MFMA peak."""
    import logging
    import torch
    import torch.optim as optim
    from pdp.factorgraph import dataset
    from pdp.trainer import SatFactorGraphTrainer
    bt = args.train_batch
    items = dataset.random_ksat_items(bt, args.n, 3, m=int(round(4.2 * args.n)), seed=555)
    b = dataset.to_torch(dataset.collate_segment(items), dev)
    gm, bvm, bfm, ef = b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature']
    label = torch.ones(bt, 1, device=dev)
    E, V = int(gm.size(1)), int(bvm.numel())
    out = {}
    for mt in ('np-nd-np', 'p-nd-np'):
        cfg = dict(model_type=mt, model_name='bench-train-' + mt, verbose=False, dropout=0.2, error_dim=3, exploration=0.1, hidden_dim=128,
                   local_search_iteration=0, epsilon=0.5, tolerance=0.02, t_max=100, edge_feature_dim=1, meta_feature_dim=0, prediction_dim=1,
                   mem_hidden_dim=100, agg_hidden_dim=100, mem_agg_hidden_dim=50, classifier_dim=50, loss_sharpness=5, randomized=True,
                   train_inner_recurrence_num=1, train_outer_recurrence_num=3, clip_norm=0.65, batch_size=bt, rng='philox', random_seed=0, init_rng='device')
        cfg['lambda'] = 0.9
        torch.manual_seed(99)
        tr = SatFactorGraphTrainer(cfg, use_cuda=True, logger=logging.getLogger('bench'))
        opt = optim.Adam(tr.get_parameter_list(), lr=1e-4, weight_decay=1e-10)
        total = np.zeros(1, dtype=np.float32)
        times = []
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            tr._train_batch(total, opt, gm, bvm, bfm, ef, None, label)
            torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
        dt = min(times[1:])
        flop = 3.0 * 3 * neural_flop_per_iteration(mt, 128, E, V)
        tf = flop / dt / 1e12
        out[mt] = dict(seconds_per_train_batch=dt, first_call_seconds=times[0], flop=flop, tflops=tf, frac_of_mfma_f32_peak=tf / MFMA_F32_PEAK_TFLOPS,
                       loss_finite=bool(np.isfinite(total).all()))
        del tr, opt
        torch.cuda.empty_cache()
    out['workload'] = ('_train_batch: %d instances of n=%d (%d edges), hidden 128, 3 outer recurrences, dropout 0.2, clipped Adam step; '
                       'flop = 3 x forward flop x recurrences' % (bt, args.n, E))
    return out


def config4_items(count, seed0=1000):
    "BASELINE configs[4]'s family (SURVEY 8(d)): k in {3,4,5} per instance, alpha_k = 0.9 x (4.27, 9.93, 21.12), n ~ U{100..500}"
    from pdp.factorgraph import dataset
    rng = np.random.RandomState(0)
    alpha = {3: 0.9 * 4.27, 4: 0.9 * 9.93, 5: 0.9 * 21.12}
    items = []
    for i in range(count):
        k = int(rng.choice([3, 4, 5])); n = int(rng.randint(100, 501))
        items += dataset.random_ksat_items(1, n, k, m=int(round(alpha[k] * n)), seed=seed0 + i)
    return items


def bench_neural(args, dev, rank, world):
    """configs[2]: fully neural PDP (np-nd-np, hidden_dim 128, layer widths 100/100/50/50) on random 3-SAT n=200.
    A step = T iterations of propagate (2 deep-set aggregators) / decimate (2 GRU cells) / predict / terminate on a resident
    batch with seeded random-init weights (the reference ships none).  SURVEY.md 8(d): 573 752 flop per edge and
    48 500 per variable and iteration, all in fp32 MFMA."""
    import torch
    from pdp import native
    from pdp.factorgraph import dataset
    T = args.iters
    m_cl = int(round(4.2 * args.n))
    items = dataset.random_ksat_items(args.batch, args.n, 3, m=m_cl, seed=1000003 * rank)
    b = dataset.to_torch(dataset.collate_segment(items), dev)
    tr, model = make_neural_model(args, T)
    E, V, F = b['graph_map'].size(1), b['batch_variable_map'].numel(), b['batch_function_map'].numel()
    iters_done, step_ms = [], []

    def step(record):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        it = neural_step(tr, model, b, T)
        torch.cuda.synchronize()
        if record:
            step_ms.append(1e3 * (time.perf_counter() - t0)); iters_done.append(it)

    for _ in range(args.warmup):
        step(False)
    if grouped():
        import torch.distributed as dist
        dist.barrier()
    native.kernel_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timing = native.kernel_timing_read(); native.kernel_timing(False)
    tot = torch.tensor([float(sum(iters_done)), elapsed], dtype=torch.float64, device=args.coll_dev)
    ranks = 1
    if grouped():
        import torch.distributed as dist
        dist.barrier()
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=args.coll_dev); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        ranks = dist.get_world_size()
    if rank == 0:
        iters_all = float(tot[0].item())
        value = iters_all / elapsed
        cpu = None if (args.no_cpu_baseline or world > 1) else cpu_baseline_neural(args)
        flops_iter = neural_flop_per_iteration('np-nd-np', args.hidden, E, V)
        achieved = flops_iter * float(np.mean(iters_done)) / (float(np.mean(step_ms)) * 1e-3) / 1e12
        print(json.dumps({
            'metric': 'pdp_iterations_per_sec', 'value': value,
            'unit': 'iterations/s (each iteration sweeps a batch of %d instances)' % args.batch, 'n_gpus': world, 'rccl_ranks': ranks, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': "configs[2]: 'np-nd-np' hidden_dim=%d (100/100/50/50), random 3-SAT n=%d m=%d batch=%d T=%d per GPU, "
                                   "seeded random-init weights" % (args.hidden, args.n, m_cl, args.batch, T),
                       'E': E, 'V': V, 'F': F, 'iterations_per_step': float(np.mean(iters_done)), 'path': model.last_run['path'],
                       'instance_iterations_per_sec': value * args.batch, 'parallelism': 'instances sharded, dp%d' % world},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / MFMA_F32_PEAK_TFLOPS,
                         'traffic': None, 'kernel': ' / '.join(native.kernel_name(k) for k in ('gru', 'agg_pre', 'agg_post')) + ' (v_mfma_f32_32x32x2_f32)',
                         'kernels': neural_kernel_rooflines(native, timing, E, V, args.hidden),
                         'note': 'achieved = neural_flop_per_iteration (573752 E + 48500 V at hidden 128) x iterations / step time (whole step, all kernels); '
                                 'kernels: HIP events of the library around every launch, flop = the MACs of that kernel x 2'},
            'cpu_baseline': cpu}))
    if grouped():
        import torch.distributed as dist
        dist.destroy_process_group()


def fast_build_measurement(args, dev, native, host_batch, items, parity_solved):
    """The headline step loop once more on the opt-in fast build (libpdp_hip_fast.so: device math on v_exp_f32 / v_log_f32 / v_rcp_f32;
    gated by the reference-held fixtures only, tests/test_fast_build_gpu.py) -- the same --warmup / --steps on the same resident batch --
    and configs[2]'s neural iteration on it.  Reported next to the line's `value`, which is always the parity build's.  Runs last:
    every handle of the parity library is gone by then (a handle belongs to the library that made it)."""
    import torch
    from pdp.factorgraph import dataset
    out = {'build': 'libpdp_hip_fast.so (PDP_BUILD=fast / pdp.native.use_build)', 'gate': 'tests/test_fast_build_gpu.py'}
    previous = native.use_build('fast')
    try:
        b = dataset.to_torch(host_batch, dev)
        prob = native.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'], batch_size=args.batch)
        E, V, F, B = prob.E, prob.V, prob.F, prob.B
        L = native.lib()
        launches, iters_done = [], []

        def step(record):
            q = torch.full((E, 3), 1.0, device=dev); q.div_(3.0)
            fs = torch.zeros(E, 2, device=dev); fs[:, 0] = 0.5
            am = torch.ones(B, dtype=torch.uint8, device=dev); dec = native.Decimator(prob)
            native.check(L.pdp_problem_bind_state(prob._h, native.ptr(prob.active_variables), native.ptr(prob.active_functions), native.ptr(prob.solution),
                                                  native.ptr(prob.is_sat), native.ptr(prob.edge_mask), native._stream()))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            prob.simplify()
            it, _ = prob.sp_solve(q, fs, am, dec, args.iters, args.tolerance, args.t_max, time_kernels=True, isolate_instances=args.isolated, inputs_disposable=True)
            torch.cuda.synchronize()
            if record:
                launches.append((time.perf_counter() - t0, dict(prob.last_solve_stats))); iters_done.append(it)
        for _ in range(args.warmup):
            step(False)
        for _ in range(args.steps):
            step(True)
        elapsed = sum(t for t, _ in launches)
        n_launch = float(np.mean([l['launches'] for _, l in launches]))
        launch_ms = float(np.mean([l['solve_kernel_ms'] for _, l in launches])) / n_launch
        bytes_launch = algorithmic_bytes_per_iteration(E, V, F) * float(np.mean(iters_done)) / n_launch
        prob.random_fill(seed=12345)
        res, _ = prob.local_search(prob.solution.clone(), args.walksat, 0.5, seed=999)
        pred = prob.update_solution(res.reshape(-1).contiguous())
        solved, unsat = prob.cnf_eval(pred.reshape(-1).contiguous())
        out.update(value=float(sum(iters_done)) / elapsed, ms_per_step=1e3 * elapsed / args.steps, kernel=native.kernel_name('sp_solve'),
                   kernel_ms_per_launch=launch_ms, roofline_frac=bytes_launch / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   solved_fraction=float(solved.sum().item()) / B, solved_fraction_parity_build=parity_solved,
                   unsat_clauses_total=float(unsat.sum().item()),
                   note='per-step time here is the host clock around bind + simplify + solve of each step (initial states built outside it)')
        del prob, b
        torch.cuda.empty_cache()
        if not args.no_secondary:
            try:
                out['neural'] = neural_shard(args, dev, native, items, 'np-nd-np', args.hidden, args.secondary_neural_iters,
                                             workload="configs[2] on the fast build: 'np-nd-np' hidden %d, same batch" % args.hidden)
            except Exception as ex:                          # measurement only: never take the headline line down
                out['neural'] = dict(error=repr(ex))
    finally:
        torch.cuda.empty_cache()
        native.use_build(previous)
    return out


def secondary_measurements(args, dev, b, prob, native, items):
    """Outside the headline's timed loop (rank 0, N = 1): the other hot kernels on the same resident batch, each with the numbers its
    roofline fraction is computed from -- configs[2]'s neural iteration (3 sweeps), 1 000 Walk-SAT steps, the Reinforce solver's forward."""
    import torch
    E, V, F, B = prob.E, prob.V, prob.F, prob.B
    out = {}
    # ---- neural: np-nd-np hidden 128, T = 3 on the same instances (configs[2]) --------------------------------------------------------------
    try:
        T = args.secondary_neural_iters
        out['neural'] = neural_shard(args, dev, native, items, 'np-nd-np', args.hidden, T,
                                     workload="configs[2]: 'np-nd-np' hidden_dim=%d on the headline batch's instances, T=%d, seeded random-init weights"
                                              % (args.hidden, T))
        out['neural']['iterations'] = sum(out['neural']['iterations_per_segment'])
        out['neural']['iterations_per_sec'] = out['neural']['segment_iterations_per_sec']
        out['neural']['flop_per_iteration'] = out['neural']['flop_per_iteration_mean']
        out['neural']['note'] = ('per-kernel ms: HIP events recorded by the library on the launch stream around every launch (pdp_kernel_timing); '
                                 'flop per launch = MACs of that kernel (SURVEY.md 8(d)) x 2; peak = fp32-input MFMA; the better of two passes after a warm-up pass')
    except Exception as ex:                                            # a secondary measurement never costs the headline line
        out['neural'] = dict(error=repr(ex))
    # ---- Walk-SAT: 1 000 steps, Philox numbers on the device ----------------------------------------------------------------------------
    try:
        steps_req = args.secondary_walksat_steps
        prob.random_fill(seed=4321)
        start = prob.solution.clone()
        prob.local_search(start, 10, 0.5, seed=5)                     # warm-up
        torch.cuda.synchronize()
        native.kernel_timing(True)
        t0 = time.perf_counter()
        res, steps = prob.local_search(start, steps_req, 0.5, seed=999)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        timing = native.kernel_timing_read(); native.kernel_timing(False)
        kms, kn = timing['walksat']
        pred = prob.update_solution(res.reshape(-1).contiguous())
        solved, unsat = prob.cnf_eval(pred.reshape(-1).contiguous())
        out['walksat'] = dict(workload='%d Walk-SAT steps (epsilon 0.5, Philox) on the headline batch from the random fill' % steps_req,
                              steps=steps, call_seconds=dt, kernel=native.kernel_name('walksat'), kernel_ms=kms, kernel_launches=kn,
                              steps_per_sec=steps / dt, flips_per_sec=steps * B / dt, us_per_step=(1e3 * kms / steps) if steps else None,
                              solved_fraction=float(solved.sum().item()) / B, unsat_clauses_total=float(unsat.sum().item()),
                              bound='latency: one workgroup per instance, a step = an LDS scan of the n variables into two 64-bit LDS arg-max atomics, '
                                    'the flip, an O(degree) integer update, two workgroup barriers; every instance of the batch is resident at once '
                                    'or in a few rounds.  The kernel is incremental and LDS-resident: the streaming model of SURVEY 8(d) (13 E + 8 V '
                                    'bytes per full re-evaluation step) does not describe it (a fraction above 1 came out of it), so no roofline '
                                    'fraction is claimed -- flips/s is the figure of merit; tools/ws_prof.py splits a step into its phases')
    except Exception as ex:
        out['walksat'] = dict(error=repr(ex))
    # ---- Reinforce solver: the persistent kernel's other instantiation --------------------------------------------------------------------
    try:
        T = args.iters
        L = native.lib()
        q = torch.empty(E, 3, device=dev); fs = torch.empty(E, 2, device=dev)
        am = torch.empty(B, dtype=torch.uint8, device=dev)
        dec = native.Decimator(prob)
        g = torch.Generator(device='cpu'); g.manual_seed(77)
        coins = torch.rand(T, generator=g).to(dev)
        runs = []
        for rep in range(3):
            native.check(L.pdp_problem_bind_state(prob._h, native.ptr(prob.active_variables), native.ptr(prob.active_functions),
                                                  native.ptr(prob.solution), native.ptr(prob.is_sat), native.ptr(prob.edge_mask), native._stream()))
            q.fill_(1.0); q.div_(3.0); fs.zero_(); fs[:, 0] = 0.5; am.fill_(1); dec.reset()
            prob.simplify()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            it, lds = prob.sp_solve(q, fs, am, dec, T, 0.01, 0.0, pi=0.1, model=native.MODEL_REINFORCE, coins=coins, decimation_probability=0.5,
                                    time_kernels=True)
            torch.cuda.synchronize()
            runs.append((time.perf_counter() - t0, it, lds, dict(prob.last_solve_stats)))
        dt, it, lds, st = runs[-1]
        per_launch = st['solve_kernel_ms'] / max(1, st['launches'])
        bytes_launch = algorithmic_bytes_per_iteration(E, V, F) * it / max(1, st['launches'])
        ach = bytes_launch / (per_launch * 1e-3) / 1e9
        out['reinforce'] = dict(workload="'reinforce' (pi 0.1, decimation probability 0.5) on the headline batch, T=%d, the persistent loop" % T,
                                iterations=it, call_seconds=dt, iterations_per_sec=it / dt, path='persistent-lds' if lds else 'persistent-hbm',
                                kernel=native.kernel_name('sp_solve'), kernel_launches=st['launches'], kernel_ms_per_launch=per_launch,
                                replay_launches=st['replays'], replay_ms=st['replay_kernel_ms'],
                                roofline=dict(bound='hbm', achieved=ach, peak=HBM_PEAK_GBS, unit='GB/s', frac=ach / HBM_PEAK_GBS,
                                              note='streaming-model bytes (41E+36V+8F per iteration) x iterations per launch / launch time'))
    except native.SpeculationFailed as ex:
        out['reinforce'] = dict(error='speculation failed: %s' % ex)
    except Exception as ex:
        out['reinforce'] = dict(error=repr(ex))
    return out


def config_shard_measurements(args, dev, native):
    """BASELINE configs[3] and configs[4] at the shape ONE GPU of the 8 gets (the 8-GPU runs deal whole loader batches to ranks, pdp/parallel.py),
    outside the headline's timed loop: configs[3] = np-nd-np hidden 128 on 5 000 instances of n=400 m=1680 (one loader batch of 40 000 / 8)
    for T sweeps + 1 000 Walk-SAT steps; configs[4] = p-nd-np hidden 128 on mixed k-SAT, batch_replication 4, the reference's dynamic
    segments (limit x hidden), T sweeps + 30 Walk-SAT steps.  T is short (the sweeps cost the same each): per-sweep rates, not solved counts."""
    from pdp.factorgraph import dataset
    out = {}
    T = args.secondary_neural_iters
    try:
        n3, b3 = 400, args.config3_batch
        items = dataset.random_ksat_items(b3, n3, 3, m=int(round(4.2 * n3)), seed=7000001)
        out['config3_shard'] = neural_shard(args, dev, native, items, 'np-nd-np', 128, T, walksat_steps=1000,
                                            workload="configs[3] per GPU: 'np-nd-np' hidden_dim=128, random 3-SAT n=%d m=%d, %d instances (one loader batch of "
                                                     "the 40 000), T=%d of 200, then 1 000 Walk-SAT steps" % (n3, int(round(4.2 * n3)), b3, T))
        del items
    except Exception as ex:
        out['config3_shard'] = dict(error=repr(ex))
    try:
        out['train'] = train_measurement(args, dev, native)
    except Exception as ex:
        out['train'] = dict(error=repr(ex))
    try:
        items = config4_items(args.config4_instances)
        out['config4_shard'] = neural_shard(args, dev, native, items, 'p-nd-np', 128, T, replication=4, limit=int(4e9), walksat_steps=30,
                                            workload="configs[4] per GPU: 'p-nd-np' hidden_dim=128, mixed random k-SAT k in {3,4,5}, n in [100,500], %d instances, "
                                                     "batch_replication 4, dynamic segments (-l 4e9), T=%d, then 30 Walk-SAT steps" % (args.config4_instances, T))
    except Exception as ex:
        out['config4_shard'] = dict(error=repr(ex))
    return out


def big_instance_measurements(args, dev, items, headline_value, native):
    """Instances past the LDS limit: (a) the headline batch plus ONE instance of n = 4 000 (50 400 edges): per-instance
    routing, the big instance as a workgroup team next to the LDS-resident pass; (b) one instance of n = 100 000 alone in its batch: the
    exact single-instance mode.  Same tolerance / t_max / T as the headline; best of three calls each."""
    import torch
    from pdp.factorgraph import dataset
    out = {}

    def run(its, reps=3):
        bb = dataset.to_torch(dataset.collate_segment(its), dev)
        hp = native.Problem(bb['graph_map'], bb['batch_variable_map'], bb['batch_function_map'], bb['edge_feature'])
        L = native.lib()
        q = torch.empty(hp.E, 3, device=dev); fs = torch.empty(hp.E, 2, device=dev); am = torch.empty(hp.B, dtype=torch.uint8, device=dev)
        dec = native.Decimator(hp)
        best = None
        for _ in range(reps):
            native.check(L.pdp_problem_bind_state(hp._h, native.ptr(hp.active_variables), native.ptr(hp.active_functions), native.ptr(hp.solution),
                                                  native.ptr(hp.is_sat), native.ptr(hp.edge_mask), native._stream()))
            q.fill_(1.0); q.div_(3.0); fs.zero_(); fs[:, 0] = 0.5; am.fill_(1); dec.reset(); hp.simplify()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            it, lds = hp.sp_solve(q, fs, am, dec, args.iters, args.tolerance, args.t_max)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = (dt, it, lds) if best is None or dt < best[0] else best
        return best, dict(hp.last_solve_stats), hp.E

    try:
        big = dataset.random_ksat_items(1, 4000, 3, m=int(round(4.2 * 4000)), seed=99)
        (dt0, it0, _), _, _ = run(items)
        (dt, it, lds), st, E = run(items + big)
        out['mixed_batch'] = dict(workload='the headline batch + one instance of n=4000 (%d edges): per-instance routing' % big[0][2].shape[1],
                                  iterations=it, call_seconds=dt, iterations_per_sec=it / dt, plain_batch_iterations_per_sec=it0 / dt0,
                                  fraction_of_plain_batch=(it / dt) / (it0 / dt0), lds_resident=bool(lds), hbm_instances=st['hbm_instances'])
    except Exception as ex:
        out['mixed_batch'] = dict(error=repr(ex))
    try:
        n1 = 100000
        one = dataset.random_ksat_items(1, n1, 3, m=int(round(3.5 * n1)), seed=11)
        (dt, it, lds), st, E = run(one)
        out['single_instance'] = dict(workload='one instance of n=%d (%d edges, alpha 3.5) alone in its batch: exact single-instance mode, one launch' % (n1, E),
                                      iterations=it, call_seconds=dt, iterations_per_sec=it / dt, edge_updates_per_sec=2.0 * E * it / dt,
                                      lds_resident=bool(lds), hbm_instances=st['hbm_instances'], kernel_launches=st['launches'])
    except Exception as ex:
        out['single_instance'] = dict(error=repr(ex))
    return out


def solved_fractions(args, dev, b, native, rank):
    """The metric's "(and solved %)": the whole forward (simplify, T sweeps, random fill, w Walk-SAT steps, Philox numbers) at the headline
    setting and at a longer one, with the reference's batch-wide semantics and with isolated instances.  Untimed."""
    import logging
    import torch
    from pdp.trainer import SatFactorGraphTrainer
    out = {}
    gm, bvm, bfm, ef = b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature']
    for name, T, w, iso in (('T%d_w%d_reference_semantics' % (args.iters, args.walksat), args.iters, args.walksat, False),
                            ('T1000_w1000_reference_semantics', 1000, 1000, False), ('T1000_w1000_isolated_instances', 1000, 1000, True)):
        try:
            tr = SatFactorGraphTrainer(dict(model_type='p-d-p', model_name='bench', verbose=False, local_search_iteration=w, epsilon=0.5,
                                            tolerance=args.tolerance, t_max=args.t_max, rng='philox', random_seed=12345 + rank, hidden_dim=3,
                                            isolated=iso, test_batch_limit=1 << 62, batch_size=args.batch, test_recurrence_num=T),
                                       use_cuda=True, logger=logging.getLogger('bench'))
            m = tr._model_list[0]
            with torch.no_grad():
                st = m.get_init_state(gm, bvm, bfm, ef, None, randomized=False, batch_replication=1)
                pred, _ = m(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                            is_training=False, iteration_num=T, check_termination=tr._check_recurrence_termination, batch_replication=1)
                solved, unsat = tr._cnf_evaluator(pred[0], gm, bvm, bfm, ef, None, sat_problem=m._last_problem)
            out[name] = dict(solved=int(solved.sum().item()), instances=int(solved.numel()), solved_fraction=float(solved.mean().item()),
                             unsat_clauses_total=float(unsat.sum().item()), iterations=m.last_run['iterations'], path=m.last_run['path'])
        except Exception as ex:
            out[name] = dict(error=repr(ex))
    # the fully neural solver with the weights this build trained on the MI355X (models/README.md): the metric's "solved %" for a neural
    # config that does not run on random weights.  Same batch, T sweeps + the same Walk-SAT budget, Philox numbers.
    wpath = os.path.join(REPO, 'models', 'demo-np-nd-np-h128.pt')
    if os.path.exists(wpath):
        name = 'np-nd-np_trained_weights_T%d_w%d' % (args.iters, args.walksat)
        try:
            cfg = dict(model_type='np-nd-np', model_name='bench-trained', verbose=False, local_search_iteration=args.walksat, epsilon=0.5, rng='philox',
                       random_seed=12345 + rank, hidden_dim=128, edge_feature_dim=1, meta_feature_dim=0, prediction_dim=1, mem_hidden_dim=100,
                       agg_hidden_dim=100, mem_agg_hidden_dim=50, classifier_dim=50, test_batch_limit=1 << 62, batch_size=args.batch,
                       test_recurrence_num=args.iters, tolerance=args.tolerance, t_max=args.t_max)
            tr = SatFactorGraphTrainer(cfg, use_cuda=True, logger=logging.getLogger('bench'))
            m = tr._model_list[0]
            m.load_state_dict(torch.load(wpath, map_location=dev), strict=True)
            with torch.no_grad():
                st = m.get_init_state(gm, bvm, bfm, ef, None, randomized=False, batch_replication=1)
                pred, _ = m(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                            is_training=False, iteration_num=args.iters, check_termination=tr._check_recurrence_termination, batch_replication=1)
                solved, unsat = tr._cnf_evaluator(pred[0], gm, bvm, bfm, ef, None, sat_problem=m._last_problem)
            out[name] = dict(solved=int(solved.sum().item()), instances=int(solved.numel()), solved_fraction=float(solved.mean().item()),
                             unsat_clauses_total=float(unsat.sum().item()), iterations=m.last_run['iterations'], path=m.last_run['path'],
                             weights='models/demo-np-nd-np-h128.pt (trained by tools/train_demo.py on n in [10, 40])')
            del tr, m
            torch.cuda.empty_cache()
        except Exception as ex:
            out[name] = dict(error=repr(ex))
    return out


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
    n_states = args.warmup + args.steps
    states = []
    for _ in range(n_states):
        q_ = torch.full((E, 3), 1.0, device=dev); q_.div_(3.0)
        fs_ = torch.zeros(E, 2, device=dev); fs_[:, 0] = 0.5
        states.append((q_, fs_, torch.ones(B, dtype=torch.uint8, device=dev), native.Decimator(prob)))
    q, fs, am, dec = states[0]
    L = native.lib()
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    kernel_ms, iters_done, paths, launches = [], [], [], []

    step_no = [0]

    def step(record):
        # a fresh SATProblem (solver.py:49-54: the library resets the problem's state arrays) + simplify(), then the solver on this step's
        # initial state
        q, fs, am, dec = states[step_no[0] % n_states]
        step_no[0] += 1
        native.check(L.pdp_problem_bind_state(prob._h, native.ptr(prob.active_variables), native.ptr(prob.active_functions),
                                              native.ptr(prob.solution), native.ptr(prob.is_sat), native.ptr(prob.edge_mask),
                                              native._stream()))
        prob.simplify()
        ev0.record()
        try:
            it, lds = prob.sp_solve(q, fs, am, dec, args.iters, args.tolerance, args.t_max, time_kernels=True, isolate_instances=args.isolated,
                                    inputs_disposable=True)      # like the solver class: this step's q / fs are copies of the initial state
            path = 'persistent-lds' if lds else 'persistent-hbm'
        except native.SpeculationFailed:
            raise SystemExit("bench: speculation failed on the benchmark batch (unexpected)")
        ev1.record()
        if record:
            torch.cuda.synchronize()
            kernel_ms.append(ev0.elapsed_time(ev1)); iters_done.append(it); paths.append(path); launches.append(dict(prob.last_solve_stats))

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
        kms = float(np.mean(kernel_ms))                 # HIP-event time of one pdp_sp_solve call (all its launches)
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
                traffic = pj.get('k_sp_solve_lds_bytes_per_launch')
                if pj.get('SQ_INSTS_VALU_per_launch'):
                    # what binds the LDS-resident kernel: wave-level VALU instructions against the issue slots of 1024 SIMDs over the measured
                    # launch time.  A wave64 fp32 instruction occupies its SIMD for 2 cycles when the same wave has an independent instruction
                    # next and ~4 in a dependent chain (tools/micro/pk_rate.hip: 2.2 / 4.3 measured; MI355X_MICROARCH.md: 2 cycles) -- both given
                    insts = float(pj['SQ_INSTS_VALU_per_launch'])
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
        line = {
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
        }
        print(json.dumps(line))
    if grouped():
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""bench.py's cpu_baseline leg: the C oracle over the host cores, the torch sparse-mm restatement of the reference, the neural oracle.
Runs on rank 0 at N = 1 only, BEFORE the process touches the GPU (the oracle workers are forked)."""
import json
import os
import sys
import time

import numpy as np

from . import REPO, HBM_PEAK_GBS, MFMA_F32_PEAK_TFLOPS, N_SIMD, CLOCK_HZ, algorithmic_bytes_per_iteration, grouped

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

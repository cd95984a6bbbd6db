#!/usr/bin/env python3
"""Benchmark of the PDP hot path on MI355X.

Metric (BASELINE.json): PDP message-passing iterations/sec on random 3-SAT n=200 m=840 batch=5000 ('p-d-p' survey
propagation + sequential decimation, T=100, configs[1]).  One "step" = one pass of the hot path over one resident
batch: reset of the solver state, SATProblem.simplify, and the T-iteration propagate/decimate/predict/terminate
loop (reference: src/pdp/nn/solver.py:332-337,355-386) -- the loop runs as ONE persistent kernel launch.
value = executed PDP iterations per second, aggregated over all ranks (each rank owns its own batch of 5000
instances: weak scaling, no collective on the data path; one RCCL all-reduce of the solved counters at the end).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank per GPU through
torch.distributed.run.  Rank 0 prints ONE JSON line.
"""

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, 'pdp-solver_amd'))

from pdp import native  # noqa: E402
from pdp.factorgraph import dataset  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_iteration(E, V, F):
    "SURVEY.md section 8(d): streaming model of one SP iteration, 41 B/edge + 36 B/variable + 8 B/clause"
    return 41 * E + 36 * V + 8 * F


def cpu_baseline(args):
    """CPU restatement (oracle, single thread) timed on a bounded sample of the same workload."""
    sys.path.insert(0, REPO)
    from oracle import binding
    binding.build()
    bs, ts = args.cpu_sample_batch, args.cpu_sample_iters
    items = dataset.random_ksat_items(bs, args.n, 3, seed=777)
    b = dataset.collate_segment(items)
    p = binding.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'])
    t0 = time.perf_counter()
    res = p.forward('p-d-p', ts, local_search_iterations=0, tolerance=args.tolerance, t_max=args.t_max, seed=1)
    dt = time.perf_counter() - t0
    inst_iters = bs * res['iterations_run'] / dt
    return dict(value=inst_iters / args.batch, unit='iterations/s (batch of %d instances)' % args.batch, cores=1, kind='port',
                sample='%d instances x %d iterations of the same n=%d m=%d 3-SAT family in %.1f s (%.0f instance-iterations/s), '
                       'scaled linearly to the batch' % (bs, res['iterations_run'], args.n, int(round(4.2 * args.n)), dt, inst_iters))


MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA = the fp32 vector peak


def cpu_baseline_neural(args):
    """The oracle's operators of one np-nd-np iteration (2 edge aggregators, 2 GRU cells, predictor aggregator + head), single thread,
    on a few instances of the same family with random weights of the same shapes; scaled linearly to the batch."""
    sys.path.insert(0, REPO)
    from oracle import binding
    binding.build()
    bs, H = 400, args.hidden
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


def bench_neural(args, dev, rank, world):
    """configs[2]: fully neural PDP (np-nd-np, hidden_dim 128, layer widths 100/100/50/50) on random 3-SAT n=200.
    A step = T iterations of propagate (2 deep-set aggregators) / decimate (2 GRU cells) / predict / terminate on a resident
    batch with seeded random-init weights (the reference ships none).  SURVEY.md 8(d): 573 752 flop per edge and
    48 500 per variable and iteration, all in fp32 MFMA."""
    import logging
    from pdp.trainer import SatFactorGraphTrainer
    T = args.iters
    m_cl = int(round(4.2 * args.n))
    items = dataset.random_ksat_items(args.batch, args.n, 3, m=m_cl, seed=1000003 * rank)
    b = dataset.to_torch(dataset.collate_segment(items), dev)
    cfg = dict(model_type='np-nd-np', model_name='bench-np-nd-np', verbose=False, local_search_iteration=0, epsilon=0.5, rng='philox',
               random_seed=1, hidden_dim=args.hidden, edge_feature_dim=1, meta_feature_dim=0, prediction_dim=1, mem_hidden_dim=100,
               agg_hidden_dim=100, mem_agg_hidden_dim=50, classifier_dim=50, test_batch_limit=1 << 62, batch_size=args.batch,
               test_recurrence_num=T)
    torch.manual_seed(1234)
    tr = SatFactorGraphTrainer(cfg, use_cuda=True, logger=logging.getLogger('bench'))
    model = tr._model_list[0]
    gm, bvm, bfm, ef = b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature']
    E, V, F = gm.size(1), bvm.numel(), bfm.numel()
    iters_done, step_ms = [], []

    def step(record):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad():
            st = model.get_init_state(gm, bvm, bfm, ef, None, randomized=False, batch_replication=1)
            model(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                  is_training=False, iteration_num=T, check_termination=tr._check_recurrence_termination, batch_replication=1)
        torch.cuda.synchronize()
        if record:
            step_ms.append(1e3 * (time.perf_counter() - t0)); iters_done.append(model.last_run['iterations'])

    for _ in range(args.warmup):
        step(False)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tot = torch.tensor([float(sum(iters_done)), elapsed], dtype=torch.float64, device=args.coll_dev)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=args.coll_dev); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    if rank == 0:
        iters_all = float(tot[0].item())
        value = iters_all / elapsed
        cpu = None if (args.no_cpu_baseline or world > 1) else cpu_baseline_neural(args)
        flops_iter = 573752.0 * E + 48500.0 * V
        achieved = flops_iter * float(np.mean(iters_done)) / (float(np.mean(step_ms)) * 1e-3) / 1e12
        print(json.dumps({
            'metric': 'pdp_iterations_per_sec', 'value': value,
            'unit': 'iterations/s (each iteration sweeps a batch of %d instances)' % args.batch, 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': "configs[2]: 'np-nd-np' hidden_dim=%d (100/100/50/50), random 3-SAT n=%d m=%d batch=%d T=%d per GPU, "
                                   "seeded random-init weights" % (args.hidden, args.n, m_cl, args.batch, T),
                       'E': E, 'V': V, 'F': F, 'iterations_per_step': float(np.mean(iters_done)), 'path': model.last_run['path'],
                       'instance_iterations_per_sec': value * args.batch, 'parallelism': 'instances sharded, dp%d' % world},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / MFMA_F32_PEAK_TFLOPS,
                         'traffic': None, 'kernel': 'k_gru_pipe / k_agg_pre_wave / k_agg_post_pf (v_mfma_f32_32x32x2_f32)',
                         'note': 'achieved = (573752 E + 48500 V) flop per iteration x iterations / step time (whole step, all kernels)'},
            'cpu_baseline': cpu}))
    if world > 1:
        import torch.distributed as dist
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
    ap.add_argument('--isolated', action='store_true', help="sp workload: every instance solved on its own (no batch-wide couplings of the "
                    "reference, hence no NaN-poison replay); the default is the reference's strict semantics")
    ap.add_argument('--seed-rank', type=int, default=None, help='generate the batch another rank would get (checks of the sharded run on one GPU)')
    ap.add_argument('--cpu-sample-batch', type=int, default=1000)
    ap.add_argument('--cpu-sample-iters', type=int, default=100)
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    backend = os.environ.get('PDP_DIST_BACKEND', 'nccl')    # gloo: several ranks on one GPU (checks of the N > 1 code on a single-GPU box)
    if world > 1:
        import torch.distributed as dist
        local_rank = local_rank % max(1, torch.cuda.device_count()) if backend != 'nccl' else local_rank
        torch.cuda.set_device(local_rank)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    native.require_gpu()
    dev = torch.device('cuda', local_rank if world > 1 else 0)
    args.coll_dev = dev if backend == 'nccl' else torch.device('cpu')
    torch.cuda.set_device(dev)
    if args.workload == 'neural':
        return bench_neural(args, dev, rank, world)

    # ---- synthetic batch, resident in HBM before the timed region ---------------------------------------------
    m = int(round(4.2 * args.n))
    items = dataset.random_ksat_items(args.batch, args.n, 3, m=m, seed=1000003 * (rank if args.seed_rank is None else args.seed_rank))
    host_batch = dataset.collate_segment(items)
    torch.cuda.synchronize(); t_setup = time.perf_counter()
    b = dataset.to_torch(host_batch, dev)                      # PCIe upload of the loader tensors
    prob = native.Problem(b['graph_map'], b['batch_variable_map'], b['batch_function_map'], b['edge_feature'], batch_size=args.batch)
    torch.cuda.synchronize(); setup_ms = 1e3 * (time.perf_counter() - t_setup)   # reported, never part of `value`
    E, V, F, B = prob.E, prob.V, prob.F, prob.B
    q = torch.empty(E, 3, device=dev); fs = torch.empty(E, 2, device=dev)
    am = torch.empty(B, dtype=torch.uint8, device=dev)
    dec = native.Decimator(prob)
    L = native.lib()
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    kernel_ms, iters_done, paths, launches = [], [], [], []

    def step(record):
        # state reset = get_init_state(randomized=False) + a fresh SATProblem (solver.py:49-54, pdp_propagate.py:233-235)
        native.check(L.pdp_problem_bind_state(prob._h, native.ptr(prob.active_variables), native.ptr(prob.active_functions),
                                              native.ptr(prob.solution), native.ptr(prob.is_sat), native.ptr(prob.edge_mask),
                                              native._stream()))
        q.fill_(1.0); q.div_(3.0); fs.zero_(); fs[:, 0] = 0.5; am.fill_(1); dec.reset()
        prob.simplify()
        ev0.record()
        try:
            it, lds = prob.sp_solve(q, fs, am, dec, args.iters, args.tolerance, args.t_max, time_kernels=True, isolate_instances=args.isolated)
            path = 'persistent-lds' if lds else 'persistent-hbm'
        except native.SpeculationFailed:
            raise SystemExit("bench: speculation failed on the benchmark batch (unexpected)")
        ev1.record()
        if record:
            torch.cuda.synchronize()
            kernel_ms.append(ev0.elapsed_time(ev1)); iters_done.append(it); paths.append(path); launches.append(dict(prob.last_solve_stats))

    def barrier():
        if world > 1:
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

    # ---- untimed: finish the forward pass once (random fill + Walk-SAT) for the solved fraction ------------------
    prob.random_fill(seed=12345 + rank)
    out, ws_steps = prob.local_search(prob.solution.clone(), args.walksat, 0.5, seed=999 + rank)
    pred = prob.update_solution(out.reshape(-1).contiguous())
    solved, unsat = prob.cnf_eval(pred.reshape(-1).contiguous())
    stats = torch.tensor([float(B), float(solved.sum().item()), float(unsat.sum().item()), elapsed, total_iters],
                         dtype=torch.float64, device=args.coll_dev)
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=args.coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)       # the only collective of the path: a 40-byte sum over xGMI
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
        # measured HBM traffic per launch (rocprofv3 PMC passes, corrected as MI355X_MICROARCH.md prescribes), if profiled
        traffic, valu = None, None
        pmc = os.path.join(REPO, 'profiles', 'r01_pmc_traffic.json')   # written by tools/summarize_profile.py
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                traffic = pj.get('k_sp_solve_lds_bytes_per_launch')
                if pj.get('SQ_INSTS_VALU_per_launch') and args.batch == 5000 and args.n == 200:
                    # what actually binds the LDS-resident kernel: wave-level VALU instructions x 4 cycles (one wave64 instruction per SIMD
                    # every 4 cycles) against the 1024 SIMDs x 2.4 GHz of the chip over the measured launch time
                    insts = float(pj['SQ_INSTS_VALU_per_launch'])
                    valu = {'insts_per_launch': insts, 'issue_frac': insts * 4.0 / (1024 * 2.4e9 * launch_ms * 1e-3),
                            'note': 'SQ_INSTS_VALU (rocprofv3 PMC pass of the same kernel) x 4 cycles / (1024 SIMDs x 2.4 GHz x launch time)'}
            except Exception:
                traffic, valu = None, None
        line = {
            'metric': 'pdp_iterations_per_sec', 'value': value,
            'unit': 'iterations/s (each iteration sweeps a batch of %d instances)' % args.batch,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': "configs[1]: 'p-d-p' survey propagation, random 3-SAT n=%d m=%d batch=%d T=%d per GPU" % (args.n, m, args.batch, args.iters),
                       'E': E, 'V': V, 'F': F, 'iterations_per_step': it_mean, 'path': paths[0] if paths else None,
                       'instance_iterations_per_sec': value * args.batch, 'edge_updates_per_sec': value * 2 * E,
                       'solve_call_ms': kms, 'kernel_launches_per_call': n_launch, 'kernel_ms_per_launch': launch_ms,
                       'poison_replay_launches_per_call': n_replay, 'poison_replay_ms_per_call': replay_ms,
                       'algorithmic_bytes_per_launch': bytes_launch, 'setup_ms_upload_and_layout': setup_ms, 'solved_fraction': n_solved / n_inst, 'unsat_clauses_total': n_unsat,
                       'walksat_steps': ws_steps, 'tolerance': args.tolerance, 't_max': args.t_max, 'parallelism': 'instances sharded, dp%d' % world,
                       'semantics': 'isolated instances' if args.isolated else "reference (batch-wide couplings reproduced)"},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'valu_issue': valu, 'kernel': 'k_sp_solve_lds<false, false, false>',
                         'note': 'achieved = streaming-model algorithmic bytes (41E+36V+8F per iteration) x iterations per launch / '
                                 'average launch duration (HIP events on the launch stream); the instance state is LDS-resident, '
                                 'so the kernel is bound by VALU issue, not by HBM (DESIGN.md section 4)'},
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""bench.py's neural measurements: configs[2] (np-nd-np, hidden 128) with per-kernel rooflines, the configs[3] / configs[4] per-GPU shards,
a training step; `bench.py --workload neural` (bench_neural)."""
import json
import os
import sys
import time

import numpy as np

from .cpu_baselines import cpu_baseline_neural
from . import REPO, HBM_PEAK_GBS, MFMA_F32_PEAK_TFLOPS, N_SIMD, CLOCK_HZ, algorithmic_bytes_per_iteration, grouped


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


_NEURAL_PMC = None


def issue_bound(key, kernel_name, ms_per_launch, units):
    """What bounds an fp32-MFMA kernel on gfx950: v_mfma_f32_32x32x2_f32 occupies the SIMD's VALU issue for its 64 cycles -- an MFMA wave beside
    a VALU wave on one SIMD takes the SUM of their times (profiles/r06_mfma_valu_overlap.txt: overlap 0.05; the bf16 MFMA overlaps 0.85) --
    so a launch cannot be shorter than (matrix-pipe busy cycles + 2 cycles x the other VALU wave-instructions) / 1024 SIMDs / 2.4 GHz.
    Instruction counts: the newest committed rocprofv3 PMC summary (profiles/*_neural_pmc.json; per launch at its size, scaled by this
    launch's edges / variables), not this run; `frac` = that floor / this run's launch time."""
    global _NEURAL_PMC
    if _NEURAL_PMC is None:
        _NEURAL_PMC = {}
        try:
            d = os.path.join(REPO, 'profiles')
            f = sorted(x for x in os.listdir(d) if x.endswith('_neural_pmc.json'))[-1]
            _NEURAL_PMC = json.load(open(os.path.join(d, f)))
            _NEURAL_PMC['file'] = 'profiles/' + f
        except (OSError, IndexError, ValueError):
            pass
    k = (_NEURAL_PMC.get('kernels') or {}).get(key)
    if not k or not kernel_name or not str(kernel_name).startswith(k['kernel'].split('<')[0] + '<') or ms_per_launch <= 0:
        return None
    scale = float(units) / float(_NEURAL_PMC['variables' if k['unit'] == 'variables' else 'edges'])
    mfma_cycles = k['SQ_VALU_MFMA_BUSY_CYCLES'] * scale
    valu = (k['SQ_INSTS_VALU'] - k['SQ_INSTS_MFMA']) * scale
    floor_ms = 1e3 * (mfma_cycles + 2.0 * valu) / (N_SIMD * CLOCK_HZ)
    out = dict(floor_ms=floor_ms, frac=floor_ms / ms_per_launch, mfma_share_of_floor=mfma_cycles / (mfma_cycles + 2.0 * valu),
               wait_share=k.get('SQ_WAIT_ANY_over_SQ_WAVE_CYCLES'), source=_NEURAL_PMC['file'])
    if k.get('clock_ghz_profiled'):
        # the fp32-MFMA kernels run power-throttled (2.1-2.3 GHz in the profiled run): the same floor at THAT clock
        out.update(clock_ghz_profiled=k['clock_ghz_profiled'], frac_at_profiled_clock=out['frac'] * CLOCK_HZ / (k['clock_ghz_profiled'] * 1e9))
    return out


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
        ib = issue_bound(key, row.get('kernel'), per, V if key == 'predict_head' else E)
        if ib:
            row['issue_bound'] = ib
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
    # The rate comes from passes WITHOUT the library's per-launch events (they cannot be read back from a replayed HIP graph: with them on, the
    # solver's device-driven loop stands back and the sweeps run one host round trip each); one more pass with the events on gives the per-kernel times.
    dt, its, timing = None, None, None
    torch.cuda.reset_peak_memory_stats()
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        its_ = [neural_step(tr, model, b, T, replication) for b in batches]
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        if dt is None or dt_ < dt:
            dt, its = dt_, its_
    loop_path = model.last_run['path']
    native.kernel_timing(True)
    t0 = time.perf_counter()
    for b in batches:
        neural_step(tr, model, b, T, replication)
    torch.cuda.synchronize()
    dt_timed = time.perf_counter() - t0
    timing = native.kernel_timing_read(); native.kernel_timing(False)
    flop = sum(neural_flop_per_iteration(model_type, hidden, e, v) * it for e, v, it in zip(E_seg, V_seg, its))
    tf = flop / dt / 1e12
    n_seg = float(len(segs))
    out = dict(workload=workload, model_type=model_type, hidden=hidden, instances=len(items), batch_replication=replication,
               segments=[len(sg) for sg in segs], edges_per_segment_with_replicas=E_seg, iterations_per_segment=its, seconds=dt,
               segment_iterations_per_sec=sum(its) / dt, ms_per_iteration_mean=1e3 * dt / max(1, sum(its)), flop_total=flop,
               flop_per_iteration_mean=flop / max(1, sum(its)), path=loop_path, seconds_with_kernel_events=dt_timed,
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
        kernels = neural_kernel_rooflines(native, timing, E, V, args.hidden)
        line = {
            'metric': 'pdp_iterations_per_sec', 'value': value,
            'unit': 'iterations/s (each iteration sweeps a batch of %d instances)' % args.batch, 'n_gpus': world, 'rccl_ranks': ranks, 'collective_backend': None, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': "configs[2]: 'np-nd-np' hidden_dim=%d (100/100/50/50), random 3-SAT n=%d m=%d batch=%d T=%d per GPU, "
                                   "seeded random-init weights" % (args.hidden, args.n, m_cl, args.batch, T),
                       'E': E, 'V': V, 'F': F, 'iterations_per_step': float(np.mean(iters_done)), 'path': model.last_run['path'],
                       'instance_iterations_per_sec': value * args.batch, 'parallelism': 'instances sharded, dp%d' % world},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / MFMA_F32_PEAK_TFLOPS,
                         'traffic': None, 'kernel': ' / '.join(native.kernel_name(k) for k in ('gru', 'agg_pre', 'agg_post')) + ' (v_mfma_f32_32x32x2_f32)',
                         'note': 'achieved = neural_flop_per_iteration (573752 E + 48500 V at hidden 128) x iterations / step time (whole step, all kernels); '
                                 'kernels: HIP events of the library around every launch, flop = the MACs of that kernel x 2'},
            'cpu_baseline': cpu}
        # per-kernel rooflines first (short detail lines), the compact record last (benchlib/line.py)
        from .line import _split, _clean, _num, _short
        details = []
        _split('roofline.kernels', _clean(kernels, 5), details)
        _split('roofline.note', line['roofline'].pop('note'), details)
        for l in details:
            print(l)
        line['config'] = {k: _num(v) for k, v in line['config'].items()}
        line['roofline'] = {k: _num(v) for k, v in line['roofline'].items()}
        for k in ('value', 'ms_per_step'):
            line[k] = _num(line[k])
        if cpu is not None:
            line['cpu_baseline'] = {k: (_short(v, 200) if isinstance(v, str) else _num(v)) for k, v in cpu.items()}
        print(json.dumps(line, allow_nan=False))
    if grouped():
        import torch.distributed as dist
        dist.destroy_process_group()

"""bench.py's measurements next to the headline line (N = 1, outside the timed loop): the opt-in fast build, configs[2] kernels, Walk-SAT,
Reinforce, the per-GPU shards of configs[3] / configs[4], big instances, solved fractions."""
import json
import os
import sys
import time

import numpy as np

from .neural import neural_shard, train_measurement, config4_items
from . import REPO, HBM_PEAK_GBS, MFMA_F32_PEAK_TFLOPS, N_SIMD, CLOCK_HZ, algorithmic_bytes_per_iteration, grouped


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
                # the fast build's GRU runs on three-term bf16 products (k_gru_bf3): the fp32-MFMA fraction above is a comparison with the
                # parity build, not a bound.  Its own bounds: the bf16 MFMA peak / 3 for the cell's flop, and the HBM stream of a call.
                nk = out['neural'].get('kernels', {}).get('gru')
                if nk and str(nk.get('kernel', '')).startswith('k_gru_bf3'):
                    E = out['neural']['edges_per_segment_with_replicas'][0]
                    stream = 3.0 * E * args.hidden * 4 + 1.0 * E * args.hidden * 4        # state + hidden in, hidden out, hidden once more for the blend (L2 / MALL)
                    sec = nk['ms_per_launch'] * 1e-3
                    nk.update(dtype='bf16x3 -> f32 (hi hi + hi lo + lo hi, fp32 accumulation)',
                              roofline_bf16x3=dict(bound='mfma', achieved=nk['tflops'], peak=2500.0 / 3.0, unit='TFLOP/s', frac=nk['tflops'] / (2500.0 / 3.0)),
                              roofline_hbm=dict(bound='hbm', achieved=stream / sec / 1e9, peak=HBM_PEAK_GBS, unit='GB/s', frac=stream / sec / 1e9 / HBM_PEAK_GBS,
                                                bytes_per_launch=stream))
                    out['neural']['dtype'] = 'f32 activations; GRU and aggregator products bf16x3 -> f32 (hi hi + hi lo + lo hi)'
                # bf16x3 kernels run on the bf16 matrix pipe: a fraction of the fp32 MFMA peak says nothing about them -- dropped here (verdict r5)
                out['neural'].pop('roofline', None)
                for row in (out['neural'].get('kernels') or {}).values():
                    if isinstance(row, dict) and (str(row.get('kernel', '')).find('bf3') >= 0 or 'frac_of_mfma_f32_peak' in row):
                        row.pop('frac_of_mfma_f32_peak', None); row.pop('issue_bound', None)
                out['neural']['note'] = ('opt-in fast build (frozen in round 6: narrower arithmetic than the reference\'s, gated by tests/test_fast_build_gpu.py): '
                                         'rates only; the GRU carries its own bounds (bf16 peak / 3, HBM stream)')
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
        # what binds it, priced like the headline kernel's valu_issue: wave-level VALU instructions of a committed rocprofv3 pass of the same
        # launch shape against the issue slots of the SIMDs over THIS run's kernel time (labelled with its source; not measured in this run)
        try:
            pm = sorted(f for f in os.listdir(os.path.join(REPO, 'profiles')) if f.endswith('_walksat_pmc.json'))
            if pm and steps_req == 1000 and B == 5000 and args.n == 200 and kms > 0:
                pj = json.load(open(os.path.join(REPO, 'profiles', pm[-1])))
                insts = float(pj['SQ_INSTS_VALU_per_launch']) * steps / float(pj['steps'])
                cyc = N_SIMD * CLOCK_HZ * kms * 1e-3
                out['walksat']['valu_issue'] = dict(insts_per_launch=insts, issue_frac_at_2_cycles=insts * 2.0 / cyc, wait_share=pj.get('SQ_WAIT_ANY_over_SQ_WAVE_CYCLES'),
                                                    source='profiles/' + pm[-1],
                                                    note='SQ_INSTS_VALU of the committed pass x 2 cycles / (1024 SIMDs x 2.4 GHz x this run\'s kernel time); the waves '
                                                         'are parked more than half of their cycles (two barriers and two LDS arg-max round trips per step): latency-bound')
        except Exception:
            pass
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


def driver_summary(config):
    """Flat scalars of everything measured next to the headline: every BASELINE config gets `<name>_<figure>` keys (iterations/s, ms per iteration, roofline fraction of
    the parity build) here; bench.py prints them as the last detail line and copies ten of them into the final line's `config`."""
    out = {}
    sec = config.get('secondary') or {}

    def put(prefix, row, keys):
        if not isinstance(row, dict) or 'error' in row:
            out[prefix + '_error'] = (row or {}).get('error', 'missing') if isinstance(row, dict) else 'missing'
            return
        for name, path in keys:
            v = row
            for p in path:
                v = v.get(p) if isinstance(v, dict) else None
            if isinstance(v, (int, float, str)) and not isinstance(v, bool):
                out[prefix + '_' + name] = v
    neural_keys = (('it_per_s', ('segment_iterations_per_sec',)), ('ms_per_iteration', ('ms_per_iteration_mean',)), ('frac_mfma_f32', ('roofline', 'frac')),
                   ('tflops', ('roofline', 'achieved')), ('reserved_gb', ('max_memory_reserved_gb',)))
    put('configs2_np_nd_np_h128', sec.get('neural'), neural_keys)
    put('configs3_shard_n400', sec.get('config3_shard'), neural_keys + (('walksat_us_per_step', ('walksat', 'us_per_step')),))
    put('configs4_shard_p_nd_np_b4', sec.get('config4_shard'), neural_keys + (('walksat_us_per_step', ('walksat', 'us_per_step')),))
    for mt in ('np-nd-np', 'p-nd-np'):
        put('train_' + mt.replace('-', '_'), (sec.get('train') or {}).get(mt), (('ms_per_batch', ('seconds_per_train_batch',)), ('frac_mfma_f32', ('frac_of_mfma_f32_peak',))))
        k = 'train_' + mt.replace('-', '_') + '_ms_per_batch'
        if k in out:
            out[k] = 1e3 * out[k]
    put('walksat_1000', sec.get('walksat'), (('flips_per_s', ('flips_per_sec',)), ('us_per_step', ('us_per_step',)), ('valu_issue_frac_at_2_cycles', ('valu_issue', 'issue_frac_at_2_cycles'))))
    put('reinforce', sec.get('reinforce'), (('it_per_s', ('iterations_per_sec',)), ('kernel_ms_per_launch', ('kernel_ms_per_launch',)), ('frac_hbm_model', ('roofline', 'frac'))))
    for name, kern in (('agg_pre', 'agg_pre'), ('agg_post', 'agg_post'), ('gru', 'gru'), ('predict_head', 'predict_head')):
        put('configs2_kernel_' + name, ((sec.get('neural') or {}).get('kernels') or {}).get(kern),
            (('ms', ('ms_per_launch',)), ('frac_mfma_f32', ('frac_of_mfma_f32_peak',)), ('frac_issue_bound', ('issue_bound', 'frac'))))
    fb = config.get('fast_build')
    if fb is not None:
        put('fast_build', fb, (('it_per_s', ('value',)), ('kernel_ms_per_launch', ('kernel_ms_per_launch',)), ('frac_hbm_model', ('roofline_frac',))))
        # the fast build's neural kernels run bf16x3 products on the bf16 matrix pipe: never a fraction of the fp32 MFMA peak
        put('fast_build_configs2', fb.get('neural') if isinstance(fb, dict) else None,
            neural_keys[:2] + (('gru_frac_of_bf16_peak_over_3', ('kernels', 'gru', 'roofline_bf16x3', 'frac')), ('gru_frac_hbm_stream', ('kernels', 'gru', 'roofline_hbm', 'frac'))))
    for name, row in (config.get('solved') or {}).items():
        if isinstance(row, dict) and 'solved' in row:
            out['solved_' + name] = '%d/%d' % (row['solved'], row['instances'])
    return out

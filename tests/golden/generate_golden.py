#!/usr/bin/env python3
"""Golden-vector generator: drives the UNMODIFIED reference (/root/reference/src) on small seeded
problems and dumps inputs + outputs as .npz/.json fixtures next to this file.

Runs ONLY in a container that has /root/reference (it never travels to the GPU box; the fixtures
do).  Nothing from the reference is copied: the reference is imported, executed, and its tensors
are recorded.  Compatibility shims for a modern stack are applied on
the oracle side only and do not change semantics:
  B-1 yaml.load default Loader, B-2 numpy legacy scalar repr for dimacs2json,
  B-3 termination check with a cloned index (same semantics, avoids self-aliased index_put),
  B-4 integer division in _deduplicate (batch replication).

Usage:  python tests/golden/generate_golden.py            # regenerates every fixture
"""

import io
import json
import logging
import os
import sys
import time
import tempfile
import warnings

import numpy as np

warnings.filterwarnings('ignore')

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference/src'

import torch  # noqa: E402
import yaml  # noqa: E402

# ---- shims (oracle side only) -------------------------------------------------------------
np.set_printoptions(legacy='1.25')                                                    # B-2
_yl = yaml.load
yaml.load = lambda s, Loader=None: _yl(s, Loader=Loader or yaml.FullLoader)           # B-1

sys.path.insert(0, REF)
import pdp.trainer as RT  # noqa: E402   (the reference)
import pdp.nn.solver as RS  # noqa: E402
import pdp.nn.util as RU  # noqa: E402
import pdp.nn.pdp_predict as RP  # noqa: E402
import pdp.nn.pdp_propagate as RPR  # noqa: E402
import pdp.nn.pdp_decimate as RD  # noqa: E402
from pdp.factorgraph.dataset import FactorGraphDataset  # noqa: E402

# our own generator lives in a package that is also called `pdp`; load it by path
import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location(
    'pdp_amd_generator', os.path.join(REPO, 'pdp-solver_amd', 'pdp', 'generator.py'))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)


def _check(self, active, prediction, sp):                                             # B-3
    out, _ = self._cnf_evaluator(variable_prediction=prediction[0], graph_map=sp._graph_map,
                                 batch_variable_map=sp._batch_variable_map,
                                 batch_function_map=sp._batch_function_map,
                                 edge_feature=sp._edge_feature, meta_data=sp._meta_data)
    idx = active[:, 0].clone().bool()
    if sp._batch_replication > 1:
        real = torch.mm(sp._replication_mask_tuple[1], (out > 0.5).float())
        dup = torch.mm(sp._replication_mask_tuple[0], (real == 0).float())
        active[idx, 0] = (dup[idx, 0] > 0).to(active.dtype)
    else:
        active[idx, 0] = (out[idx, 0] <= 0.5).to(active.dtype)


RT.SatFactorGraphTrainer._check_recurrence_termination = _check


def _dedup(self, prediction, propagator_state, decimator_state, sp):                  # B-4
    if sp._batch_replication <= 1 or sp._replication_mask_tuple is None:
        return None, None, None
    assignment = 2 * prediction[0] - 1.0
    energy, _ = self._compute_energy(assignment, sp)
    max_ind = RU.sparse_argmax(-energy.squeeze(1), sp._replication_mask_tuple[0], device=self._device)
    batch_flag = torch.zeros(sp._batch_size, 1, device=self._device)
    batch_flag[max_ind, 0] = 1
    flag = torch.mm(sp._batch_mask_tuple[0], batch_flag)
    variable_prediction = (flag * prediction[0]).view(sp._batch_replication, -1).sum(dim=0).unsqueeze(1)
    flag = torch.mm(sp._graph_mask_tuple[1], flag)
    r, e = sp._batch_replication, sp._edge_num // sp._batch_replication
    new_p = tuple((flag * x).view(r, e, -1).sum(dim=0) for x in propagator_state) \
        if propagator_state is not None else None
    new_d = tuple((flag * x).view(r, e, -1).sum(dim=0) for x in decimator_state) \
        if decimator_state is not None else None
    return (variable_prediction, None), new_p, new_d


RS.PropagatorDecimatorSolverBase._deduplicate = _dedup

LOG = logging.getLogger('golden')

# ---- recording torch.rand -------------------------------------------------------------------
_TORCH_RAND = torch.rand        # the real function, saved once: what every `finally` restores
_real_rand = torch.rand         # what the recorder calls (rf_leak feeds it from a file for one run)
RAND_LOG = []


def _rec_rand(*a, **k):
    out = _real_rand(*a, **k)
    RAND_LOG.append(out.detach().clone().reshape(-1).numpy())
    return out


def np_(t):
    return None if t is None else t.detach().cpu().numpy().copy()


# ---- problems -------------------------------------------------------------------------------

def make_lines(specs, seed0):
    """specs: list of (n, m, k_choices). Returns JSON lines (compact format)."""
    lines = []
    for i, (n, m, ks) in enumerate(specs):
        rng = np.random.RandomState(seed0 + i)
        clauses = []
        for _ in range(m):
            k = int(ks[rng.randint(0, len(ks))])
            variables = rng.choice(n, size=min(k, n), replace=False) + 1
            signs = rng.randint(0, 2, size=len(variables)) * 2 - 1
            clauses.append([int(v * s) for v, s in zip(variables, signs)])
        lines.append(gen.json_line(n, clauses, label=i % 2, name="g%d" % i))
    return lines


def collate(lines, limit=10 ** 12, hidden_dim=1):
    """Run the reference loader (parse + collate) on JSON lines -> list of torch tensors."""
    with tempfile.NamedTemporaryFile('w', suffix='.json', delete=False) as f:
        f.write("\n".join(lines) + "\n")
        path = f.name
    ds = FactorGraphDataset(path, limit=limit, hidden_dim=hidden_dim)
    items = [ds._convert_line(l) for l in lines]
    out = ds.dag_collate_fn(items)
    os.unlink(path)
    return out


def batch_tensors(lines):
    gm, bvm, bfm, ef, gf, lab, misc = collate(lines)
    assert len(gm) == 1
    return gm[0], bvm[0], bfm[0], ef[0], lab[0], misc[0]


def base_cfg(model_type, **kw):
    cfg = dict(model_type=model_type, model_name='golden-' + model_type, verbose=False, dropout=0,
               error_dim=1, exploration=0, hidden_dim=3, local_search_iteration=0, epsilon=0.5,
               tolerance=0.02, t_max=100, pi=0.01, decimation_probability=0.5,
               edge_feature_dim=1, meta_feature_dim=0, prediction_dim=1, mem_hidden_dim=100,
               agg_hidden_dim=100, mem_agg_hidden_dim=50, classifier_dim=50, loss_sharpness=5)
    cfg.update(kw)
    return cfg


def build(cfg, seed=1234):
    torch.manual_seed(seed)
    tr = RT.SatFactorGraphTrainer(cfg, use_cuda=False, logger=LOG)
    return tr, tr._model_list[0]


def problem_arrays(gm, bvm, bfm, ef):
    return dict(graph_map=np_(gm).astype(np.int32), batch_variable_map=np_(bvm).astype(np.int32),
                batch_function_map=np_(bfm).astype(np.int32), edge_feature=np_(ef).astype(np.float32))


def save(name, **arrs):
    arrs = {k: v for k, v in arrs.items() if v is not None}
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024.0))


# ---- A. loader + problem set-up + simplify ---------------------------------------------------

MIXED_SPECS = [(20, 60, (3,)), (12, 30, (2, 3)), (20, 85, (3,)), (15, 40, (1, 2, 3)), (25, 60, (3, 4)),
               (10, 42, (3,)), (18, 50, (2, 3, 4, 5)), (20, 84, (3,))]


def gen_loader_and_simplify():
    lines = make_lines(MIXED_SPECS, seed0=100)
    gm, bvm, bfm, ef, lab, misc = batch_tensors(lines)
    out = problem_arrays(gm, bvm, bfm, ef)
    out['label'] = np_(lab)
    with open(os.path.join(HERE, 'mixed_batch.jsonl'), 'w') as f:
        f.write("\n".join(lines) + "\n")
    dev = torch.device('cpu')
    sp = RS.SATProblem((gm, bvm, bfm, ef, None, None), dev, 1)
    sp.simplify()
    out['simplify_active_variables'] = np_(sp._active_variables[:, 0])
    out['simplify_active_functions'] = np_(sp._active_functions[:, 0])
    out['simplify_solution'] = np_(sp._solution)
    out['simplify_is_sat'] = np_(sp._is_sat)
    # set_variables with a hand-made partial assignment on the simplified problem
    rng = np.random.RandomState(5)
    assign = np.zeros((bvm.numel(), 1), dtype=np.float32)
    pick = rng.choice(bvm.numel(), size=12, replace=False)
    assign[pick, 0] = rng.randint(0, 2, size=12) * 2 - 1
    out['setvar_assignment'] = assign[:, 0].copy()
    sp.set_variables(torch.from_numpy(assign.copy()))
    out['setvar_active_variables'] = np_(sp._active_variables[:, 0])
    out['setvar_active_functions'] = np_(sp._active_functions[:, 0])
    out['setvar_solution'] = np_(sp._solution)
    out['setvar_is_sat'] = np_(sp._is_sat)
    # replication map sanity (a18): replicated problem arrays
    sp3 = RS.SATProblem((gm, bvm, bfm, ef, None, None), dev, 3)
    out['rep3_graph_map'] = np_(sp3._graph_map).astype(np.int32)
    out['rep3_batch_variable_map'] = np_(sp3._batch_variable_map).astype(np.int32)
    out['rep3_batch_function_map'] = np_(sp3._batch_function_map).astype(np.int32)
    out['rep3_edge_feature'] = np_(sp3._edge_feature)
    save('problem_simplify', **out)
    return lines


# ---- B. per-operator vectors (K-groups) ------------------------------------------------------

def gen_ops(lines):
    gm, bvm, bfm, ef, lab, misc = batch_tensors(lines)
    dev = torch.device('cpu')
    E, V, F = gm.size(1), bvm.numel(), bfm.numel()
    B = int(bvm.max()) + 1
    out = problem_arrays(gm, bvm, bfm, ef)
    rng = np.random.RandomState(11)

    sp = RS.SATProblem((gm, bvm, bfm, ef, None, None), dev, 1)
    sp.simplify()
    # de-activate a few more things by fixing variables so masks are non-trivial
    assign = torch.zeros(V, 1)
    pick = rng.choice(V, size=10, replace=False)
    assign[pick, 0] = torch.from_numpy((rng.randint(0, 2, size=10) * 2 - 1).astype(np.float32))
    sp.set_variables(assign)
    out['active_variables'] = np_(sp._active_variables[:, 0])
    out['active_functions'] = np_(sp._active_functions[:, 0])
    out['solution'] = np_(sp._solution)
    edge_mask = torch.mm(sp._graph_mask_tuple[1], sp._active_variables) * \
        torch.mm(sp._graph_mask_tuple[3], sp._active_functions)
    sp._edge_mask = edge_mask
    out['edge_mask'] = np_(edge_mask[:, 0])

    # --- smooth max / max / argmax (K4, K5)
    x_e = torch.from_numpy(rng.rand(E, 1).astype(np.float32))
    x_e[rng.rand(E) < 0.1] = 0
    out['smax_in'] = np_(x_e[:, 0])
    out['smax_out'] = np_(RU.sparse_smooth_max(x_e, sp._graph_mask_tuple[0], dev)[:, 0])
    x_v = torch.from_numpy((rng.rand(V).astype(np.float32) * 0.5))
    x_v[rng.rand(V) < 0.2] = 0
    x_v[3] = x_v[5] = x_v.max()  # ties
    out['vmax_in'] = np_(x_v)
    out['vmax_out'] = np_(RU.sparse_max(x_v, sp._batch_mask_tuple[0], dev))
    out['vargmax_out'] = np_(RU.sparse_argmax(x_v, sp._batch_mask_tuple[0], dev)).astype(np.int64)
    x_neg = torch.from_numpy((rng.randn(V).astype(np.float32)))
    out['vmax_neg_in'] = np_(x_neg)
    out['vmax_neg_out'] = np_(RU.sparse_max(x_neg, sp._batch_mask_tuple[0], dev))
    out['vargmax_neg_out'] = np_(RU.sparse_argmax(x_neg, sp._batch_mask_tuple[0], dev)).astype(np.int64)

    # --- CNF evaluator (K9)
    ev = RU.SatCNFEvaluator(dev)
    pred = torch.from_numpy(rng.rand(V, 1).astype(np.float32))
    pred[rng.rand(V) < 0.3] = 0.5
    pred[rng.rand(V) < 0.2] = 1.0
    pred[rng.rand(V) < 0.2] = 0.0
    solved, unsat = ev(pred, gm, bvm, bfm, ef, None)
    out['cnf_pred'] = np_(pred[:, 0]); out['cnf_solved'] = np_(solved[:, 0]); out['cnf_unsat'] = np_(unsat[:, 0])

    # --- Survey propagator (K1-K3) with and without masks, pi = 0 and pi = 0.1
    for tag, pi in (('pi0', 0.0), ('pi1', 0.1)):
        prop = RPR.SurveyPropagator(dev, decimator_dimension=1, include_adaptors=False, pi=pi)
        q = torch.from_numpy(rng.rand(E, 3).astype(np.float32)); q = q / q.sum(1, keepdim=True)
        q[rng.rand(E) < 0.05, 0] = 0.0          # exercise the 1e-40 clamp
        fs = torch.from_numpy(rng.rand(E, 2).astype(np.float32))
        fs[rng.rand(E) < 0.05, 0] = 1.0         # 1 - eta == 0 -> clamp
        fs[:, 1] = torch.from_numpy(rng.randint(-1, 2, size=E).astype(np.float32)) if pi > 0 else 0
        init_q = torch.from_numpy(rng.rand(E, 3).astype(np.float32))
        init_fs = torch.from_numpy(rng.rand(E, 2).astype(np.float32))
        am = torch.ones(B, 1, dtype=torch.uint8); am[1] = 0; am[6] = 0
        out['sp_%s_q' % tag] = np_(q); out['sp_%s_fs' % tag] = np_(fs)
        out['sp_%s_init_q' % tag] = np_(init_q); out['sp_%s_init_fs' % tag] = np_(init_fs)
        out['sp_%s_active_mask' % tag] = np_(am[:, 0])
        o1 = prop((init_q, init_fs), (q, fs, edge_mask), sp, False, am)
        out['sp_%s_masked_q' % tag] = np_(o1[0]); out['sp_%s_masked_fs' % tag] = np_(o1[1])
        o2 = prop((init_q, init_fs), (q, fs), sp, False, None)
        out['sp_%s_plain_q' % tag] = np_(o2[0]); out['sp_%s_plain_fs' % tag] = np_(o2[1])
        # scorer (K6)
        sc = RP.SurveyScorer(dev, message_dimension=1, include_adaptors=False, pi=pi)
        s, _ = sc((q, fs), sp)
        out['score_%s' % tag] = np_(s[:, 0])

    # --- energy / delta energy (K14 pieces)
    base = RS.PropagatorDecimatorSolverBase(dev, 'x', None, None, RP.IdentityPredictor(dev, True), 5, 0.5)
    a = torch.from_numpy((rng.randint(0, 2, size=(V, 1)) * 2 - 1).astype(np.float32)) * sp._active_variables
    en, uf = base._compute_energy(a, sp)
    out['energy_assignment'] = np_(a[:, 0]); out['energy_per_instance'] = np_(en[:, 0]); out['energy_unsat_functions'] = np_(uf[:, 0])
    out['energy_delta'] = np_(base._compute_energy_diff(a, sp)[:, 0])
    save('ops_classical', **out)


def gen_scorer_adaptors(lines):
    """SurveyScorer(include_adaptors=True) (pdp_predict.py:145-152, 161-166): a bias-free projector [H -> 2] in front of the score, column 0
    through a sigmoid (the survey), column 1 through sign (the external force).  No solver of the reference's factory builds it; the class
    takes the flag, so the plug-in API keeps it.  Fixture: message state [E, 16], projector weight, the assignment that shapes the masks,
    scores for pi = 0 and 0.1."""
    gm, bvm, bfm, ef, lab, misc = batch_tensors(lines)
    dev = torch.device('cpu')
    E, V = gm.size(1), bvm.numel()
    H = 16
    out = problem_arrays(gm, bvm, bfm, ef)
    rng = np.random.RandomState(23)
    sp = RS.SATProblem((gm, bvm, bfm, ef, None, None), dev, 1)
    sp.simplify()
    assign = torch.zeros(V, 1)
    pick = rng.choice(V, size=10, replace=False)
    assign[pick, 0] = torch.from_numpy((rng.randint(0, 2, size=10) * 2 - 1).astype(np.float32))
    sp.set_variables(assign)
    out['assign'] = np_(assign[:, 0])
    out['active_functions'] = np_(sp._active_functions[:, 0])
    msg = torch.from_numpy(rng.randn(E, H).astype(np.float32))
    out['message'] = np_(msg)
    for tag, pi in (('pi0', 0.0), ('pi01', 0.1)):
        torch.manual_seed(5)
        sc = RP.SurveyScorer(dev, message_dimension=H, include_adaptors=True, pi=pi)
        out['weight'] = np_(sc._projector.weight)
        with torch.no_grad():
            s, _ = sc((None, msg.clone()), sp)
        out['score_' + tag] = np_(s[:, 0])
    save('scorer_adaptors', **out)


def gen_predictor_function_branch(lines):
    """NeuralPredictor with BOTH classifiers (pdp_predict.py:49-91): the variable branch every solver of the factory uses and the function
    branch none of them builds (function_classifier is None in solver.py:534,558).  Small widths; weights, the two decimator states, an edge
    mask and both predictions."""
    gm, bvm, bfm, ef, lab, misc = batch_tensors(lines)
    dev = torch.device('cpu')
    E = gm.size(1)
    H = 8
    out = problem_arrays(gm, bvm, bfm, ef)
    rng = np.random.RandomState(29)
    sp = RS.SATProblem((gm, bvm, bfm, ef, None, None), dev, 1)
    torch.manual_seed(9)
    pr = RP.NeuralPredictor(dev, H, 1, 1, 0, 12, 10, 6, variable_classifier=RT.Perceptron(H, 5, 1), function_classifier=RT.Perceptron(H, 7, 1))
    for k, v in pr.state_dict().items():
        if '_module_list' not in k:
            out['w__' + k.replace('.', '__')] = np_(v)
    dv = torch.from_numpy((rng.randn(E, H) * 0.7).astype(np.float32)); df = torch.from_numpy((rng.randn(E, H) * 0.7).astype(np.float32))
    em = torch.from_numpy((rng.rand(E, 1) > 0.2).astype(np.float32))
    out['dec_v'] = np_(dv); out['dec_f'] = np_(df); out['edge_mask'] = np_(em[:, 0])
    with torch.no_grad():
        pv, pf = pr((dv, df, em), sp)
        pv2, pf2 = pr((dv, df), sp)
    out['pred_v_masked'] = np_(pv[:, 0]); out['pred_f_masked'] = np_(pf[:, 0])
    out['pred_v'] = np_(pv2[:, 0]); out['pred_f'] = np_(pf2[:, 0])
    save('predictor_function_branch', **out)


# ---- C. traces of the full solver ------------------------------------------------------------

def run_trace(model_type, lines, T, w, seed, replication=1, cfg_kw=None, float_iters=(0, 1, 2, 5, 10),
              randomized=False, tag=None):
    cfg = base_cfg(model_type, local_search_iteration=w, **(cfg_kw or {}))
    tr, m = build(cfg)
    gm, bvm, bfm, ef, lab, misc = batch_tensors(lines)
    out = problem_arrays(gm, bvm, bfm, ef)
    out['meta'] = np.array([T, w, seed, replication], dtype=np.int64)
    torch.manual_seed(seed)
    np.random.seed(seed)
    del RAND_LOG[:]
    torch.rand = _rec_rand
    try:
        with torch.no_grad():
            st = m.get_init_state(gm, bvm, bfm, ef, None, randomized=randomized, batch_replication=replication)
            it = {'i': 0}
            ints = {k: [] for k in ('active_variables', 'active_functions', 'solution', 'active_mask')}
            floats = {}
            orig_check = tr._check_recurrence_termination

            def check(active, prediction, sp):
                orig_check(active, prediction, sp)
                ints['active_variables'].append(np_(sp._active_variables[:, 0]))
                ints['active_functions'].append(np_(sp._active_functions[:, 0]))
                ints['solution'].append(np_(sp._solution))
                ints['active_mask'].append(np_(active[:, 0]))
                it['i'] += 1

            prop_states = []

            def prop_hook(mod, inp, outp):
                i = it['i']
                if i in float_iters or i == T - 1:
                    floats['prop_q_%d' % i] = np_(outp[0])
                    floats['prop_fs_%d' % i] = np_(outp[1])

            h = None
            if m._propagator is not None:
                h = m._propagator.register_forward_hook(prop_hook)
            pred, (ps, dsn) = m(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm,
                                edge_feature=ef, meta_data=None, is_training=False, iteration_num=T,
                                check_termination=check, batch_replication=replication)
            if h is not None:
                h.remove()
    finally:
        torch.rand = _TORCH_RAND
    out['iterations_run'] = np.array([it['i']], dtype=np.int64)
    for k, v in ints.items():
        if v:
            out['trace_' + k] = np.stack(v)
    out.update(floats)
    out['final_prediction'] = np_(pred[0][:, 0])
    if ps is not None:
        for j, x in enumerate(ps):
            out['final_prop_%d' % j] = np_(x)
        for j, x in enumerate(dsn):
            out['final_dec_%d' % j] = np_(x)
    ev = RU.SatCNFEvaluator(torch.device('cpu'))
    if replication == 1:
        solved, unsat = ev(pred[0], gm, bvm, bfm, ef, None)
        out['final_solved'] = np_(solved[:, 0]); out['final_unsat'] = np_(unsat[:, 0])
    out['rand_sizes'] = np.array([len(r) for r in RAND_LOG], dtype=np.int64)
    out['rand_stream'] = np.concatenate(RAND_LOG).astype(np.float32) if RAND_LOG else np.zeros(0, np.float32)
    save(tag or ('trace_' + model_type), **out)
    return m, tr


def gen_traces():
    specs50 = [(50, 210, (3,))] * 8
    lines50 = make_lines(specs50, seed0=500)
    with open(os.path.join(HERE, 'sat50_batch.jsonl'), 'w') as f:
        f.write("\n".join(lines50) + "\n")
    # easier instances (alpha = 3.5) so that decimation + walksat actually solve some
    easy = make_lines([(40, 140, (3,))] * 10, seed0=900)
    with open(os.path.join(HERE, 'sat40_easy_batch.jsonl'), 'w') as f:
        f.write("\n".join(easy) + "\n")
    run_trace('p-d-p', lines50, T=40, w=0, seed=7, tag='trace_pdp_n50')
    run_trace('p-d-p', easy, T=60, w=30, seed=7, tag='trace_pdp_easy_ws',
              cfg_kw=dict(tolerance=0.05, t_max=10))
    run_trace('p-d-p', make_lines(MIXED_SPECS, seed0=100), T=30, w=10, seed=3, tag='trace_pdp_mixed',
              cfg_kw=dict(tolerance=0.05, t_max=8))
    run_trace('walk-sat', easy, T=50, w=50, seed=11, tag='trace_walksat_easy')
    run_trace('p-d-p', easy[:4], T=25, w=20, seed=5, replication=3, tag='trace_pdp_rep3',
              cfg_kw=dict(tolerance=0.05, t_max=6))
    run_trace('reinforce', easy, T=40, w=10, seed=13, tag='trace_reinforce_easy')
    # random initial state (the test mode's, base.py:229): the first sweep reads the decimator's random state
    run_trace('p-d-p', make_lines([(30, 108, (3,))] * 10 + [(24, 60, (2, 3, 4))] * 6, seed0=4300), T=25, w=15, seed=21, randomized=True,
              tag='trace_pdp_randinit', cfg_kw=dict(tolerance=0.05, t_max=10))


# ---- D. neural operators + traces ------------------------------------------------------------

def flat_state_dict(m):
    """Canonical (non-aliased) tensors only; the alias map is stored separately as JSON."""
    sd = m.state_dict()
    keep = {}
    for k, v in sd.items():
        if k.startswith(('_propagator.', '_decimator.', '_predictor.')) and '_module_list' not in k:
            keep['w__' + k.replace('.', '__')] = np_(v)
    return keep


def alias_map(m):
    """Every state-dict key -> the canonical key holding the same storage."""
    sd = m.state_dict()
    canon = {}
    for k, v in sd.items():
        if k.startswith(('_propagator.', '_decimator.', '_predictor.')) and '_module_list' not in k:
            canon[v.data_ptr()] = k
    return {k: canon.get(v.data_ptr(), k) for k, v in sd.items()}


def gen_neural():
    big = make_lines([(20, 70, (3,))] * 5 + [(14, 40, (2, 3, 4))], seed0=1300)
    small = make_lines([(12, 36, (3,)), (10, 30, (2, 3, 4)), (12, 40, (3,))], seed0=1400)
    with open(os.path.join(HERE, 'neural_batch.jsonl'), 'w') as f:
        f.write("\n".join(big) + "\n")
    with open(os.path.join(HERE, 'neural_batch_small.jsonl'), 'w') as f:
        f.write("\n".join(small) + "\n")
    for tag, H, lines, T in (('h32', 32, big, 5), ('h128', 128, small, 2)):
        gm, bvm, bfm, ef, lab, misc = batch_tensors(lines)
        cfg = base_cfg('np-nd-np', hidden_dim=H, local_search_iteration=0)
        tr, m = build(cfg, seed=1234)
        out = problem_arrays(gm, bvm, bfm, ef)
        out.update(flat_state_dict(m))
        if H == 32:
            with open(os.path.join(HERE, 'state_dict_alias_map.json'), 'w') as f:
                json.dump(alias_map(m), f, indent=0, sort_keys=True)
        keep_iters = (0, 1, T - 1)
        it = {'i': 0}
        rec = {}
        orig_check = tr._check_recurrence_termination

        def check(active, prediction, sp):
            i = it['i']
            rec['pred_%d' % i] = np_(prediction[0][:, 0])
            orig_check(active, prediction, sp)
            rec['active_mask_%d' % i] = np_(active[:, 0])
            it['i'] += 1

        def prop_hook(mod, inp, outp):
            if it['i'] in keep_iters:
                rec['prop_v_%d' % it['i']] = np_(outp[0]); rec['prop_f_%d' % it['i']] = np_(outp[1])

        def dec_hook(mod, inp, outp):
            if it['i'] in keep_iters:
                rec['dec_v_%d' % it['i']] = np_(outp[0]); rec['dec_f_%d' % it['i']] = np_(outp[1])

        h1 = m._propagator.register_forward_hook(prop_hook)
        h2 = m._decimator.register_forward_hook(dec_hook)
        torch.manual_seed(3)
        with torch.no_grad():
            st = m.get_init_state(gm, bvm, bfm, ef, None, randomized=True, batch_replication=1)
            out['init_prop_v'] = np_(st[0][0]); out['init_prop_f'] = np_(st[0][1])
            out['init_dec_v'] = np_(st[1][0]); out['init_dec_f'] = np_(st[1][1])
            pred, _ = m(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm,
                        edge_feature=ef, meta_data=None, is_training=False, iteration_num=T,
                        check_termination=check, batch_replication=1)
        h1.remove(); h2.remove()
        out.update(rec)
        out['final_prediction'] = np_(pred[0][:, 0])
        out['meta'] = np.array([T, H], dtype=np.int64)
        save('trace_neural_' + tag, **out)


def gen_np_d_np():
    "model type np-d-np: neural propagator + sequential decimator scored by a neural predictor (solver.py:616-637)"
    lines = make_lines([(16, 52, (3,))] * 6, seed0=2100)
    with open(os.path.join(HERE, 'npdnp_batch.jsonl'), 'w') as f:
        f.write("\n".join(lines) + "\n")
    gm, bvm, bfm, ef, lab, misc = batch_tensors(lines)
    H, T, w = 24, 14, 8
    cfg = base_cfg('np-d-np', hidden_dim=H, local_search_iteration=w, tolerance=0.2, t_max=3)
    tr, m = build(cfg, seed=4321)
    out = problem_arrays(gm, bvm, bfm, ef)
    for k, v in m.state_dict().items():
        if k.startswith(('_propagator.', '_decimator.', '_predictor.')) and '_module_list' not in k:
            out['w__' + k.replace('.', '__')] = np_(v)
    with open(os.path.join(HERE, 'state_dict_alias_map_npdnp.json'), 'w') as f:
        json.dump(alias_map(m), f, indent=0, sort_keys=True)
    ints = {k: [] for k in ('active_variables', 'solution', 'active_mask')}
    orig_check = tr._check_recurrence_termination

    def check(active, prediction, sp):
        orig_check(active, prediction, sp)
        ints['active_variables'].append(np_(sp._active_variables[:, 0])); ints['solution'].append(np_(sp._solution))
        ints['active_mask'].append(np_(active[:, 0]))

    torch.manual_seed(9)
    del RAND_LOG[:]
    torch.rand = _rec_rand
    try:
        with torch.no_grad():
            st = m.get_init_state(gm, bvm, bfm, ef, None, randomized=True, batch_replication=1)
            out['init_prop_v'] = np_(st[0][0]); out['init_prop_f'] = np_(st[0][1])
            pred, (ps, ds) = m(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef,
                               meta_data=None, is_training=False, iteration_num=T, check_termination=check, batch_replication=1)
    finally:
        torch.rand = _TORCH_RAND
    for k, v in ints.items():
        out['trace_' + k] = np.stack(v)
    out['final_prediction'] = np_(pred[0][:, 0])
    out['final_prop_0'] = np_(ps[0]); out['final_prop_1'] = np_(ps[1])
    out['rand_sizes'] = np.array([len(r) for r in RAND_LOG], dtype=np.int64)
    out['meta'] = np.array([T, H, w, 9], dtype=np.int64)
    save('trace_np_d_np', **out)



def gen_p_nd_np():
    """model type p-nd-np: SP propagator with learned adaptors + neural decimator + neural predictor (solver.py:543-561).
    The reference cannot construct a runnable model (NeuralDecimator((3, 1), ...) against a propagator that emits
    [eta, force]); the one-word shim below gives the decimator the function-message width 2 the propagator really has.  Everything
    else is the unmodified reference."""
    import pdp.nn.pdp_decimate as ref_dec
    orig_init = ref_dec.NeuralDecimator.__init__

    def shim_init(self, device, message_dimension, *a, **k):                  # B-5
        orig_init(self, device, (3, 2) if message_dimension == (3, 1) else message_dimension, *a, **k)

    lines = make_lines([(18, 60, (3,))] * 4 + [(14, 40, (2, 3, 4))] * 2, seed0=3100)
    with open(os.path.join(HERE, 'pndnp_batch.jsonl'), 'w') as f:
        f.write("\n".join(lines) + "\n")
    gm, bvm, bfm, ef, lab, misc = batch_tensors(lines)
    H, T = 32, 6
    cfg = base_cfg('p-nd-np', hidden_dim=H, local_search_iteration=0)
    ref_dec.NeuralDecimator.__init__ = shim_init
    try:
        tr, m = build(cfg, seed=2468)
    finally:
        ref_dec.NeuralDecimator.__init__ = orig_init
    out = problem_arrays(gm, bvm, bfm, ef)
    out.update(flat_state_dict(m))
    with open(os.path.join(HERE, 'state_dict_alias_map_pndnp.json'), 'w') as f:
        json.dump(alias_map(m), f, indent=0, sort_keys=True)
    it = {'i': 0}
    rec = {}
    orig_check = tr._check_recurrence_termination

    def check(active, prediction, sp):
        i = it['i']
        rec['pred_%d' % i] = np_(prediction[0][:, 0])
        orig_check(active, prediction, sp)
        rec['active_mask_%d' % i] = np_(active[:, 0])
        it['i'] += 1

    def prop_hook(mod, inp, outp):
        rec['prop_q_%d' % it['i']] = np_(outp[0]); rec['prop_fs_%d' % it['i']] = np_(outp[1])

    def dec_hook(mod, inp, outp):
        rec['dec_v_%d' % it['i']] = np_(outp[0]); rec['dec_f_%d' % it['i']] = np_(outp[1])

    h1 = m._propagator.register_forward_hook(prop_hook)
    h2 = m._decimator.register_forward_hook(dec_hook)
    torch.manual_seed(5)
    with torch.no_grad():
        st = m.get_init_state(gm, bvm, bfm, ef, None, randomized=True, batch_replication=1)
        out['init_prop_q'] = np_(st[0][0]); out['init_prop_fs'] = np_(st[0][1])
        out['init_dec_v'] = np_(st[1][0]); out['init_dec_f'] = np_(st[1][1])
        pred, _ = m(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm,
                    edge_feature=ef, meta_data=None, is_training=False, iteration_num=T,
                    check_termination=check, batch_replication=1)
    h1.remove(); h2.remove()
    out.update(rec)
    out['final_prediction'] = np_(pred[0][:, 0])
    out['meta'] = np.array([T, H, it['i']], dtype=np.int64)
    save('trace_p_nd_np', **out)



def gen_test_metrics():
    """test mode (satyr-train-test.py -t; base.py:223-250, trainer.py:108-123): accuracy / recall / energy loss of given predictions
    on a labelled batch, computed by the reference's own _compute_evaluation_metrics."""
    lines = make_lines([(20, 70, (3,))] * 6 + [(14, 40, (2, 3, 4))] * 3, seed0=4200)
    gm, bvm, bfm, ef, lab, misc = batch_tensors(lines)
    cfg = base_cfg('p-d-p', exploration=0.3, loss_sharpness=5)
    tr, m = build(cfg, seed=1)
    out = problem_arrays(gm, bvm, bfm, ef)
    rng = np.random.RandomState(11)
    B = int(bvm.max().item()) + 1
    label = torch.from_numpy(rng.randint(0, 2, size=(B, 1)).astype(np.float32))
    out['label'] = np_(label)
    V = bvm.numel()
    preds = [rng.rand(V).astype(np.float32), (rng.rand(V) > 0.5).astype(np.float32), (0.5 + 0.01 * rng.randn(V)).astype(np.float32)]
    steps = [1.0, 40.0, 5.0e4]
    res = []
    for k, (pr, gs) in enumerate(zip(preds, steps)):
        m._global_step.data = torch.tensor([gs], dtype=torch.float32)
        with torch.no_grad():
            met = tr._compute_evaluation_metrics(model=m, evaluator=tr._evaluator, prediction=(torch.from_numpy(pr).reshape(-1, 1), None),
                                                 label=label, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm,
                                                 edge_feature=ef, meta_data=None)
        out['pred_%d' % k] = pr
        res.append(np_(met).reshape(-1))
    out['metrics'] = np.stack(res).astype(np.float32)          # rows: [accuracy error, recall error, loss]
    out['global_step'] = np.array(steps, dtype=np.float32)
    out['params'] = np.array([0.3, 10.0, 1e-8, 5.0], dtype=np.float64)   # exploration, max_coeff, eps, loss_sharpness
    # whole test() call of the reference on a labelled JSON file (p-d-p + Walk-SAT, random initial state from the seeded torch stream)
    tlines = make_lines([(40, 140, (3,))] * 10 + [(24, 60, (2, 3, 4))] * 6, seed0=4300)       # alpha = 3.5: SP converges, Walk-SAT solves some
    with open(os.path.join(HERE, 'test_mode_batch.json'), 'w') as f:
        f.write("\n".join(tlines) + "\n")
    cfg2 = base_cfg('p-d-p', error_dim=3, exploration=0.3, loss_sharpness=5, test_recurrence_num=25, local_search_iteration=15,
                    batch_size=16, test_batch_limit=40000000, max_cache_size=100000, tolerance=0.05, t_max=10)
    tr2, m2 = build(cfg2, seed=1)
    tr2._num_cores = 0                                                    # in-process loader (same random stream, no worker processes)
    m2._global_step.data = torch.tensor([7.0], dtype=torch.float32)
    torch.manual_seed(21)
    res = tr2.test(os.path.join(HERE, 'test_mode_batch.json'), batch_replication=1)
    out['test_mode_error'] = np.asarray(res[0][1], dtype=np.float32)     # [3, 1]: accuracy error, recall error, loss
    out['test_mode_meta'] = np.array([25, 15, 16, 21, 7], dtype=np.int64)  # T, walk-sat steps, batch size, torch seed, global step
    save('test_metrics', **out)



def gen_headline_poison():
    """BASELINE configs[1]'s instance family at a batch the reference can run: uniform 3-SAT n=200 m=840, B=50, T=100, p-d-p.
    At the SAT threshold a 0/0 in the SP update (pdp_propagate.py:215-216) turns every batch-global reduction into NaN:
    from that sweep on no instance of the batch is gated or decimated.  The instances are regenerated from seeds by the build's own
    generator (the same generator bench.py uses), so only the outcome is stored."""
    # instances of bench.py's rank-0 batch (instance i = RandomState(i)); 2499, 2776, 3533 and 4730 are the ones whose surveys become NaN
    # within 100 sweeps on the MI355X run of the full 5 000-instance batch (the first at sweep 81)
    seeds = [2499, 2776, 3533, 4730] + list(range(100, 146))
    lines = []
    for sd in seeds:
        variables, signs = gen.uniform_ksat_arrays(200, 840, 3, np.random.RandomState(sd))
        vn, fn, gmap, efeat = gen.compact_arrays(200, variables, signs)
        lines.append(gen.format_json_line(vn, fn, ((gmap[0] + 1) * efeat).astype(np.int64), gmap[1] + 1, -1, "h%d" % sd))
    cfg = base_cfg('p-d-p', local_search_iteration=0, tolerance=0.02, t_max=100)
    tr, m = build(cfg)
    gm, bvm, bfm, ef, lab, misc = batch_tensors(lines)
    T = 100
    counts, nan_iter = [], []
    orig_check = tr._check_recurrence_termination

    def check(active, prediction, sp):
        orig_check(active, prediction, sp)
        counts.append(int(sp._active_variables.sum().item()))

    def prop_hook(mod, inp, outp):
        if not nan_iter and bool(torch.isnan(outp[0]).any()):
            nan_iter.append(len(counts))

    h = m._propagator.register_forward_hook(prop_hook)
    torch.manual_seed(7)
    with torch.no_grad():
        st = m.get_init_state(gm, bvm, bfm, ef, None, randomized=False, batch_replication=1)
        pred, (ps, ds) = m(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                           is_training=False, iteration_num=T, check_termination=check, batch_replication=1)
    h.remove()
    ev = RU.SatCNFEvaluator(torch.device('cpu'))
    solved, unsat = ev(pred[0], gm, bvm, bfm, ef, None)
    save('headline_n200_poison', seeds=np.array(seeds, dtype=np.int64), meta=np.array([200, 840, T, 7, len(counts)], dtype=np.int64),
         active_variable_count=np.array(counts, dtype=np.int64), first_nan_sweep=np.array(nan_iter or [-1], dtype=np.int64),
         final_bits=np.packbits((np_(pred[0][:, 0]) > 0.5).astype(np.uint8)), variable_num=np.array([int(bvm.numel())], dtype=np.int64),
         final_solved=np_(solved[:, 0]), final_unsat=np_(unsat[:, 0]))



def gen_headline_mid(B=800, w=100):
    """The headline family at a batch between the toy fixture above and the benchmark's 5 000: the four NaN-producing instances + the first
    B - 4 instances of bench.py's rank-0 batch, T = 100 sweeps, then the random fill and `w` Walk-SAT steps of the full forward (solver.py:324-353,
    433-467) on the reference's own torch.rand stream (torch.manual_seed(7)).  Minutes of CPU: the reference densifies [V x B] per reduction."""
    seeds = [2499, 2776, 3533, 4730] + list(range(0, B - 4))
    lines = _headline_lines(seeds)
    cfg = base_cfg('p-d-p', local_search_iteration=w, epsilon=0.5, tolerance=0.02, t_max=100)
    tr, m = build(cfg)
    gm, bvm, bfm, ef, lab, misc = batch_tensors(lines)
    T = 100
    counts, nan_iter = [], []
    orig_check = tr._check_recurrence_termination

    def check(active, prediction, sp):
        orig_check(active, prediction, sp)
        counts.append(int(sp._active_variables.sum().item()))

    def prop_hook(mod, inp, outp):
        if not nan_iter and bool(torch.isnan(outp[0]).any()):
            nan_iter.append(len(counts))

    h = m._propagator.register_forward_hook(prop_hook)
    torch.manual_seed(7)
    t0 = time.time()
    with torch.no_grad():
        st = m.get_init_state(gm, bvm, bfm, ef, None, randomized=False, batch_replication=1)
        pred, (ps, ds) = m(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                           is_training=False, iteration_num=T, check_termination=check, batch_replication=1)
    h.remove()
    ev = RU.SatCNFEvaluator(torch.device('cpu'))
    solved, unsat = ev(pred[0], gm, bvm, bfm, ef, None)
    print('headline_mid: B=%d, %d sweeps, first NaN %s, solved %d, unsat clauses %d, %.0f s' % (B, len(counts), nan_iter, int(solved.sum()), int(unsat.sum()), time.time() - t0))
    save('headline_n200_mid', seeds=np.array(seeds, dtype=np.int64), meta=np.array([200, 840, T, 7, len(counts), w], dtype=np.int64),
         active_variable_count=np.array(counts, dtype=np.int64), first_nan_sweep=np.array(nan_iter or [-1], dtype=np.int64),
         final_bits=np.packbits((np_(pred[0][:, 0]) > 0.5).astype(np.uint8)), variable_num=np.array([int(bvm.numel())], dtype=np.int64),
         final_solved=np_(solved[:, 0]).astype(np.uint8), final_unsat=np_(unsat[:, 0]).astype(np.int32))


def gen_headline_neural():
    """np-nd-np, hidden 128 (configs[2]) on 6 instances of bench.py's family (n=200 m=840): per-sweep predictions of the reference with the
    seeded weights of gen_neural's h128 model (same constructor seed, so trace_neural_h128.npz holds these tensors)."""
    seeds = list(range(100, 106))
    lines = []
    for sd in seeds:
        variables, signs = gen.uniform_ksat_arrays(200, 840, 3, np.random.RandomState(sd))
        vn, fn, gmap, efeat = gen.compact_arrays(200, variables, signs)
        lines.append(gen.format_json_line(vn, fn, ((gmap[0] + 1) * efeat).astype(np.int64), gmap[1] + 1, -1, "h%d" % sd))
    gm, bvm, bfm, ef, lab, misc = batch_tensors(lines)
    H, T = 128, 4
    cfg = base_cfg('np-nd-np', hidden_dim=H, local_search_iteration=0)
    tr, m = build(cfg, seed=1234)
    ref_w = np.load(os.path.join(HERE, 'trace_neural_h128.npz'))
    for k, v in flat_state_dict(m).items():
        assert np.array_equal(ref_w[k], v), k
    rec = {}
    it = {'i': 0}
    orig_check = tr._check_recurrence_termination

    def check(active, prediction, sp):
        rec['pred_%d' % it['i']] = np_(prediction[0][:, 0])
        orig_check(active, prediction, sp)
        it['i'] += 1

    torch.manual_seed(3)
    with torch.no_grad():
        st = m.get_init_state(gm, bvm, bfm, ef, None, randomized=False, batch_replication=1)      # the predict path's initial state (zeros)
        pred, (ps, ds) = m(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                           is_training=False, iteration_num=T, check_termination=check, batch_replication=1)
    rec['final_prediction'] = np_(pred[0][:, 0])
    rec['final_dec_f_sample'] = np_(ds[1][::97])
    save('headline_n200_neural', seeds=np.array(seeds, dtype=np.int64), meta=np.array([200, 840, T, H, it['i']], dtype=np.int64), **rec)



# ---- F. BASELINE configs[0] / configs[4] / long neural pins ---------------------------------------------------------------

def gen_config0_cli():
    """BASELINE configs[0]: 'p-d-p' on 100 random 3-SAT DIMACS files n=50 m=210, --cpu_mode, batch_size=100, T=50, through the reference's
    own CLI (satyr.py -d -c -z 100).  The files are regenerated from seeds by the test (generator.write_dimacs, instance i =
    RandomState(9000 + i)); stored: the reference's output rows, whose ID column also is the converter's os.listdir order."""
    import runpy
    import shutil
    ddir = tempfile.mkdtemp(prefix='cfg0_')
    try:
        for i in range(100):
            gen.write_dimacs(os.path.join(ddir, 'c0_%03d_%d.cnf' % (i, i % 2)), 50, gen.uniform_ksat(50, 210, 3, np.random.RandomState(9000 + i)))
        out_path = os.path.join(HERE, 'cli_config0.out.jsonl')
        argv = ['satyr.py', '/root/reference/config/Predict/PDP-p-d-p-sp-pytorch.yaml', ddir, '50', '-d', '-c', '-z', '100', '-s', '7', '-o', out_path]
        old_argv, old_cwd = sys.argv, os.getcwd()
        sys.argv = argv
        try:
            runpy.run_path(os.path.join(REF, 'satyr.py'), run_name='__main__')
        finally:
            sys.argv = old_argv
            os.chdir(old_cwd)
    finally:
        shutil.rmtree(ddir, ignore_errors=True)
    print('wrote', out_path)


def mixed_k_lines(count, n_lo, n_hi, seed0, alpha_scale=0.9):
    "configs[4]'s family at test size: k drawn per instance from {3,4,5}, n ~ U{n_lo..n_hi}, m = round(0.9 * alpha_k * n) (SURVEY.md 8(d))"
    alpha = {3: 4.27, 4: 9.93, 5: 21.12}
    lines, meta = [], []
    for i in range(count):
        rng = np.random.RandomState(seed0 + i)
        k = int(rng.choice([3, 4, 5])); n = int(rng.randint(n_lo, n_hi + 1))
        m = int(round(alpha_scale * alpha[k] * n))
        lines.append(gen.json_line(n, gen.uniform_ksat(n, m, k, rng), label=i % 2, name="mk%d" % i))
        meta.append((k, n, m))
    return lines, meta


def _with_b5_shim(fn):
    import pdp.nn.pdp_decimate as ref_dec
    orig_init = ref_dec.NeuralDecimator.__init__

    def shim_init(self, device, message_dimension, *a, **k):                  # B-5
        orig_init(self, device, (3, 2) if message_dimension == (3, 1) else message_dimension, *a, **k)

    ref_dec.NeuralDecimator.__init__ = shim_init
    try:
        return fn()
    finally:
        ref_dec.NeuralDecimator.__init__ = orig_init


def gen_config4_mixed():
    """BASELINE configs[4] at a size the reference runs in seconds: 'p-nd-np' (reference + the App. B-4 / B-5 shims) on mixed random k-SAT,
    k in {3,4,5}, n in [20,60], batch_replication 4, a test_batch_limit that cuts the loader batch into several dynamic segments, Walk-SAT
    with the torch random stream -- through the reference's own predict() (loader, DynamicBatchDivider, _predict_batch, _deduplicate,
    post-processor).  Stored: weights, the JSON input, per-segment per-sweep predictions, and the output rows."""
    H, T, w, R, seed = 32, 8, 25, 4, 17
    lines, meta = mixed_k_lines(14, 20, 60, seed0=8800)
    with open(os.path.join(HERE, 'config4_mixed.json'), 'w') as f:
        f.write("\n".join(lines) + "\n")
    edges = [len(json.loads(l)[1]) for l in lines]
    limit = 4 * H * (sum(edges) // 3)                 # limit // R // (max_edges * H) < batch: three or four segments, largest instances first
    cfg = base_cfg('p-nd-np', hidden_dim=H, local_search_iteration=w, test_recurrence_num=T, batch_size=14, test_batch_limit=limit,
                   max_cache_size=100000)
    tr, m = _with_b5_shim(lambda: build(cfg, seed=97531))
    tr._num_cores = 0
    out = flat_state_dict(m)
    with open(os.path.join(HERE, 'state_dict_alias_map_config4.json'), 'w') as f:
        json.dump(alias_map(m), f, indent=0, sort_keys=True)
    rec, seg = {}, {'i': -1, 'sweep': 0, 'sizes': []}
    orig_check = tr._check_recurrence_termination
    orig_batch = tr._predict_batch

    def check(active, prediction, sp):
        rec['seg%d_pred_%d' % (seg['i'], seg['sweep'])] = np_(prediction[0][:, 0])
        orig_check(active, prediction, sp)
        rec['seg%d_active_%d' % (seg['i'], seg['sweep'])] = np_(active[:, 0])
        seg['sweep'] += 1

    def predict_batch(graph_map, *a, **k):
        seg['i'] += 1; seg['sweep'] = 0
        seg['sizes'].append(int(graph_map.size(1)))
        return orig_batch(graph_map, *a, **k)

    tr._check_recurrence_termination = check
    tr._predict_batch = predict_batch
    torch.manual_seed(seed); np.random.seed(seed)
    del RAND_LOG[:]
    torch.rand = _rec_rand
    buf = io.StringIO()
    try:
        tr.predict(os.path.join(HERE, 'config4_mixed.json'), buf, import_path_base=None, post_processor=tr._post_process_predictions,
                   batch_replication=R)
    finally:
        torch.rand = _TORCH_RAND
    rows = [l for l in buf.getvalue().split('\n') if l.strip()]
    with open(os.path.join(HERE, 'config4_mixed.out.jsonl'), 'w') as f:
        f.write("\n".join(rows) + "\n")
    out.update(rec)
    out['meta'] = np.array([T, H, w, R, seed, limit, len(seg['sizes'])], dtype=np.int64)
    out['segment_edges'] = np.array(seg['sizes'], dtype=np.int64)
    out['instance_knm'] = np.array(meta, dtype=np.int64)
    out['rand_sizes'] = np.array([len(r) for r in RAND_LOG], dtype=np.int64)
    save('config4_mixed', **out)
    print('config4: %d segments (edges %s), %d rows, solved %d' % (len(seg['sizes']), seg['sizes'], len(rows),
                                                                  sum(json.loads(r)['solved'] for r in rows)))


def _headline_lines(seeds):
    lines = []
    for sd in seeds:
        variables, signs = gen.uniform_ksat_arrays(200, 840, 3, np.random.RandomState(sd))
        vn, fn, gmap, efeat = gen.compact_arrays(200, variables, signs)
        lines.append(gen.format_json_line(vn, fn, ((gmap[0] + 1) * efeat).astype(np.int64), gmap[1] + 1, -1, "h%d" % sd))
    return lines


def gen_neural_long():
    """Longer reference pins of the neural model types at configs[2]'s width: hidden 128, 8 instances of bench.py's family (n=200 m=840),
    T = 24 sweeps from the test mode's random initial state (torch seed 3).  np-nd-np uses the seeded weights of trace_neural_h128.npz (same constructor seed);
    p-nd-np (reference + App. B-5 shim) stores its own.  Stored per sweep: the prediction (all variables), the active mask, a sample of the
    decimator state; at the end the final prediction."""
    seeds = list(range(200, 208))
    lines = _headline_lines(seeds)
    gm, bvm, bfm, ef, lab, misc = batch_tensors(lines)
    H, T = 128, 24
    for mt in ('np-nd-np', 'p-nd-np'):
        cfg = base_cfg(mt, hidden_dim=H, local_search_iteration=0)
        if mt == 'np-nd-np':
            tr, m = build(cfg, seed=1234)
            ref_w = np.load(os.path.join(HERE, 'trace_neural_h128.npz'))
            for k, v in flat_state_dict(m).items():
                assert np.array_equal(ref_w[k], v), k
            rec = {}
        else:
            tr, m = _with_b5_shim(lambda: build(cfg, seed=2468))
            rec = flat_state_dict(m)
            with open(os.path.join(HERE, 'state_dict_alias_map_pndnp_h128.json'), 'w') as f:
                json.dump(alias_map(m), f, indent=0, sort_keys=True)
        it = {'i': 0}
        orig_check = tr._check_recurrence_termination

        def check(active, prediction, sp):
            rec['pred_%d' % it['i']] = np_(prediction[0][:, 0])
            orig_check(active, prediction, sp)
            rec['active_mask_%d' % it['i']] = np_(active[:, 0])
            it['i'] += 1

        def dec_hook(mod, inp, outp):
            rec['dec_v_sample_%d' % it['i']] = np_(outp[0][::997]); rec['dec_f_sample_%d' % it['i']] = np_(outp[1][::997])

        h2 = m._decimator.register_forward_hook(dec_hook)
        torch.manual_seed(3)
        with torch.no_grad():
            # the test mode's random initial state (base.py:229): regenerated by the build from the same torch seed, a sample is stored
            st = m.get_init_state(gm, bvm, bfm, ef, None, randomized=True, batch_replication=1)
            for a_, nm in ((st[0][0], 'init_prop_0'), (st[0][1], 'init_prop_1'), (st[1][0], 'init_dec_0'), (st[1][1], 'init_dec_1')):
                rec[nm + '_sample'] = np_(a_[::997])
            pred, (ps, ds) = m(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                               is_training=False, iteration_num=T, check_termination=check, batch_replication=1)
        h2.remove()
        rec['final_prediction'] = np_(pred[0][:, 0])
        save('neural_long_' + mt.replace('-', '_'), seeds=np.array(seeds, dtype=np.int64), meta=np.array([200, 840, T, H, it['i']], dtype=np.int64), **rec)



# ---- G. training path (f3) ---------------------------------------------------------------------------------------------------------

def train_cfg(**kw):
    cfg = base_cfg('np-nd-np', hidden_dim=32, error_dim=3, exploration=0.1, loss_sharpness=5, dropout=0.0, randomized=True,
                   train_inner_recurrence_num=1, train_outer_recurrence_num=3, clip_norm=0.65, batch_size=6, epoch_num=2, repetition_num=1,
                   train_batch_limit=4000000, test_batch_limit=40000000, max_cache_size=100000, test_recurrence_num=5, local_search_iteration=0)
    cfg['lambda'] = 0.9
    cfg.update(kw)
    return cfg


def gen_train():
    """The training path of the reference (FactorGraphTrainerBase._train_batch / train, base.py:149-182, 311-404) on np-nd-np, hidden 32:
    (a) one batch: the loss, the gradient of every parameter after loss.backward() (before clipping), and the parameters after the
    clipped Adam step; (b) a two-epoch train() call with dropout 0.2 on a 12-instance file (shuffled loader, validation pass):
    losses and validation errors per epoch and the final parameters."""
    import torch.optim as optim
    lines = make_lines([(18, 60, (3,))] * 4 + [(14, 40, (2, 3, 4))] * 2, seed0=6100)
    gm, bvm, bfm, ef, lab, misc = batch_tensors(lines)
    cfg = train_cfg()
    tr, m = build(cfg, seed=777)
    m._global_step.data = torch.tensor([3.0])
    out = problem_arrays(gm, bvm, bfm, ef)
    out['label'] = np_(lab)
    for k, v in flat_state_dict(m).items():
        out[k] = v
    with open(os.path.join(HERE, 'state_dict_alias_map_train.json'), 'w') as f:
        json.dump(alias_map(m), f, indent=0, sort_keys=True)
    # manual backward (same statements as _train_batch) for the raw gradients
    torch.manual_seed(31)
    lam = torch.tensor([cfg['lambda']], dtype=torch.float32)
    state = m.get_init_state(gm, bvm, bfm, ef, None, cfg['randomized'])
    loss = torch.zeros(1)
    step_losses = []
    for t in range(cfg['train_outer_recurrence_num']):
        prediction, state = m(init_state=state, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                              is_training=True, iteration_num=cfg['train_inner_recurrence_num'])
        lt = tr._compute_loss(model=m, loss=tr._loss, prediction=prediction, label=lab, graph_map=gm, batch_variable_map=bvm,
                              batch_function_map=bfm, edge_feature=ef, meta_data=None)
        step_losses.append(float(lt))
        if t == 0:
            out['first_prediction'] = np_(prediction[0][:, 0])
        loss = loss + lt * lam.pow(float(cfg['train_outer_recurrence_num'] - t - 1))
    loss.backward()
    out['loss'] = np.array([float(loss)], dtype=np.float32)
    out['step_losses'] = np.array(step_losses, dtype=np.float32)
    for name, prm in m.named_parameters(remove_duplicate=False):          # every alias is listed: keep the canonical names
        if prm.grad is not None and '_module_list' not in name and name.startswith(('_propagator.', '_decimator.', '_predictor.')):
            out['g__' + name.replace('.', '__')] = np_(prm.grad)
    # the real _train_batch on a fresh copy of the same model: parameters after the clipped Adam step
    tr2, m2 = build(cfg, seed=777)
    m2._global_step.data = torch.tensor([3.0])
    opt = optim.Adam(tr2.get_parameter_list(), lr=1e-3, weight_decay=1e-10)
    total = np.zeros(1, dtype=np.float32)
    torch.manual_seed(31)
    tr2._train_batch(total, opt, gm, bvm, bfm, ef, None, lab)
    out['train_batch_total_loss'] = total.copy()
    for k, v in flat_state_dict(m2).items():
        out['after__' + k] = v
    save('train_batch', **out)

    # (b) train(): two epochs, dropout, shuffled loader
    tl = make_lines([(16, 50, (3,))] * 8 + [(12, 30, (2, 3, 4))] * 4, seed0=6200)
    with open(os.path.join(HERE, 'train_small.json'), 'w') as f:
        f.write("\n".join(tl) + "\n")
    cfg3 = train_cfg(dropout=0.2, max_cache_size=1)      # the reference's collate adds the batch offsets INTO the cached item arrays (dataset.py:172-173):
                                                         # a cached item is wrong from its second use on; a one-item cache never serves a stale one here
    tr3, m3 = build(cfg3, seed=778)
    tr3._num_cores = 0
    opt3 = optim.Adam(tr3.get_parameter_list(), lr=1e-3, weight_decay=1e-10)
    torch.manual_seed(41); np.random.seed(41)
    path = os.path.join(HERE, 'train_small.json')
    res = {}
    for k, v in flat_state_dict(m3).items():
        res[k] = v
    _, errors, losses = tr3.train([path], [path], opt3, last_export_path_base=None, best_export_path_base=None, metric_index=0)
    res['errors'] = errors; res['losses'] = losses
    res['global_step'] = np_(m3._global_step)
    for k, v in flat_state_dict(m3).items():
        res['after__' + k] = v
    res['next_rand'] = torch.rand(4).numpy()
    save('train_run', **res)
    print('train_batch loss', float(loss), 'step losses', step_losses, '| train(): losses', losses.reshape(-1), 'errors', errors[:, 0, :, 0])



def gen_train_hybrid():
    """Training of the two hybrid model types the reference ships Train configs for (config/Train/p-prodec2-nsp-..., p-prodec2-ndec-...),
    through the same statements as _train_batch (base.py:149-182), hidden 32, the batch of gen_train:
    * p-nd-np (SP propagator with adaptors + GRU decimator + neural predictor; reference + the App. B-5 width shim, meta_feature_dim 0 --
      the shipped config's meta_feature_dim 1 builds layers one column wider than the loader's data, which never carries meta data):
      loss, per-step losses, every parameter gradient, parameters after the clipped Adam step -> train_batch_p_nd_np.npz;
    * np-d-np (neural propagator + sequential decimator with a neural scorer + identity predictor): the prediction is sat_problem._solution,
      which only depends on parameters through values a decimation wrote, and set_variables edits the flag tensors in place that the
      scorer's graph saved -- loss.backward() RAISES in the reference.  The exception is the fixture (train_np_d_np_reference.json)."""
    import torch.optim as optim
    lines = make_lines([(18, 60, (3,))] * 4 + [(14, 40, (2, 3, 4))] * 2, seed0=6100)
    gm, bvm, bfm, ef, lab, misc = batch_tensors(lines)

    def forward_loss(tr, m, cfg, out=None):
        lam = torch.tensor([cfg['lambda']], dtype=torch.float32)
        state = m.get_init_state(gm, bvm, bfm, ef, None, cfg['randomized'])
        loss = torch.zeros(1)
        step_losses, req = [], []
        for t in range(cfg['train_outer_recurrence_num']):
            prediction, state = m(init_state=state, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                                  is_training=True, iteration_num=cfg['train_inner_recurrence_num'])
            lt = tr._compute_loss(model=m, loss=tr._loss, prediction=prediction, label=lab, graph_map=gm, batch_variable_map=bvm,
                                  batch_function_map=bfm, edge_feature=ef, meta_data=None)
            step_losses.append(float(lt)); req.append(bool(prediction[0].requires_grad))
            if t == 0 and out is not None:
                out['first_prediction'] = np_(prediction[0][:, 0])
            loss = loss + lt * lam.pow(float(cfg['train_outer_recurrence_num'] - t - 1))
        return loss, step_losses, req

    def p_nd_np():
        cfg = train_cfg(model_type='p-nd-np', model_name='golden-p-nd-np')
        tr, m = build(cfg, seed=777)
        m._global_step.data = torch.tensor([3.0])
        out = problem_arrays(gm, bvm, bfm, ef)
        out['label'] = np_(lab)
        for k, v in flat_state_dict(m).items():
            out[k] = v
        with open(os.path.join(HERE, 'state_dict_alias_map_train_pndnp.json'), 'w') as f:
            json.dump(alias_map(m), f, indent=0, sort_keys=True)
        torch.manual_seed(31)
        loss, step_losses, _ = forward_loss(tr, m, cfg, out)
        loss.backward()
        out['loss'] = np.array([float(loss)], dtype=np.float32)
        out['step_losses'] = np.array(step_losses, dtype=np.float32)
        for name, prm in m.named_parameters(remove_duplicate=False):
            if prm.grad is not None and '_module_list' not in name and name.startswith(('_propagator.', '_decimator.', '_predictor.')):
                out['g__' + name.replace('.', '__')] = np_(prm.grad)
        tr2, m2 = build(cfg, seed=777)
        m2._global_step.data = torch.tensor([3.0])
        opt = optim.Adam(tr2.get_parameter_list(), lr=1e-3, weight_decay=1e-10)
        total = np.zeros(1, dtype=np.float32)
        torch.manual_seed(31)
        tr2._train_batch(total, opt, gm, bvm, bfm, ef, None, lab)
        out['train_batch_total_loss'] = total.copy()
        for k, v in flat_state_dict(m2).items():
            out['after__' + k] = v
        save('train_batch_p_nd_np', **out)
        print('p-nd-np train_batch loss', float(loss), 'step losses', step_losses, 'gradients', len([k for k in out if k.startswith('g__')]))
    _with_b5_shim(p_nd_np)

    cfg = train_cfg(model_type='np-d-np', model_name='golden-np-d-np', tolerance=0.02, t_max=10, train_outer_recurrence_num=10)    # the shipped config's values
    tr, m = build(cfg, seed=777)
    torch.manual_seed(31)
    rec = dict(model_type='np-d-np', statements='FactorGraphTrainerBase._train_batch (base.py:149-182)', torch=torch.__version__)
    try:
        loss, step_losses, req = forward_loss(tr, m, cfg)
        rec['prediction_requires_grad_per_step'] = req
        loss.backward()
        rec['raised'] = None
    except RuntimeError as ex:
        rec['raised'] = type(ex).__name__
        rec['message'] = str(ex).split('\n')[0][:200]
    with open(os.path.join(HERE, 'train_np_d_np_reference.json'), 'w') as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print('np-d-np:', rec)


def det_weights(name, shape):
    "parameter values both sides can rebuild from the canonical parameter name alone (no weight tensors in the fixture): U(-1, 1) / sqrt(fan_in)"
    import zlib
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff)
    scale = 1.0 / np.sqrt(shape[-1]) if len(shape) >= 2 else 0.1
    return (rs.uniform(-1.0, 1.0, size=shape) * scale).astype(np.float32)


def sample_index(name, numel, count=384):
    import zlib
    rs = np.random.RandomState((zlib.crc32(name.encode()) ^ 0x5bd1e995) & 0x7fffffff)
    return np.sort(rs.choice(numel, size=min(count, numel), replace=False)).astype(np.int64)


def gen_train_h128():
    """_train_batch's statements (base.py:149-182) at the hidden width the shipped configs train (hidden_dim 128 region; gen_train uses 32):
    np-nd-np and p-nd-np on a batch of 8 instances, 3 outer recurrences.  The fixture carries no weight tensors: every parameter is
    det_weights(canonical name, shape), which the test rebuilds; of every parameter gradient it keeps 384 sampled entries (sample_index),
    the largest magnitude and the L2 norm -> train_h128_<model type>.npz (a few tens of KB each)."""
    lines = make_lines([(40, 160, (3,))] * 6 + [(30, 100, (2, 3, 4))] * 2, seed0=6500)
    gm, bvm, bfm, ef, lab, misc = batch_tensors(lines)

    def run(model_type):
        cfg = train_cfg(model_type=model_type, model_name='golden-h128-' + model_type, hidden_dim=128)
        tr, m = build(cfg, seed=779)
        m._global_step.data = torch.tensor([3.0])
        amap = alias_map(m)
        with torch.no_grad():
            for k, v in m.state_dict().items():
                if k.startswith(('_propagator.', '_decimator.', '_predictor.')) and '_module_list' not in k and amap[k] == k:
                    v.copy_(torch.from_numpy(det_weights(k, tuple(v.shape))))
        out = problem_arrays(gm, bvm, bfm, ef)
        out['label'] = np_(lab)
        torch.manual_seed(37)
        lam = torch.tensor([cfg['lambda']], dtype=torch.float32)
        state = m.get_init_state(gm, bvm, bfm, ef, None, cfg['randomized'])
        loss = torch.zeros(1)
        step_losses = []
        for t in range(cfg['train_outer_recurrence_num']):
            prediction, state = m(init_state=state, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=None,
                                  is_training=True, iteration_num=cfg['train_inner_recurrence_num'])
            lt = tr._compute_loss(model=m, loss=tr._loss, prediction=prediction, label=lab, graph_map=gm, batch_variable_map=bvm,
                                  batch_function_map=bfm, edge_feature=ef, meta_data=None)
            step_losses.append(float(lt))
            if t == 0:
                out['first_prediction'] = np_(prediction[0][:, 0])
            loss = loss + lt * lam.pow(float(cfg['train_outer_recurrence_num'] - t - 1))
        loss.backward()
        out['loss'] = np.array([float(loss)], dtype=np.float32)
        out['step_losses'] = np.array(step_losses, dtype=np.float32)
        names = []
        for name, prm in m.named_parameters(remove_duplicate=False):
            if prm.grad is not None and '_module_list' not in name and name.startswith(('_propagator.', '_decimator.', '_predictor.')) and amap[name] == name:
                g = np_(prm.grad).reshape(-1)
                key = name.replace('.', '__')
                out['gs__' + key] = g[sample_index(name, g.size)]
                out['gn__' + key] = np.array([np.abs(g).max(), np.sqrt((g.astype(np.float64) ** 2).sum())], dtype=np.float64)
                names.append(name)
        ref_map = 'state_dict_alias_map_train.json' if model_type == 'np-nd-np' else 'state_dict_alias_map_train_pndnp.json'
        assert json.load(open(os.path.join(HERE, ref_map))) == amap          # parameter names do not depend on the hidden width: the test reuses that map
        save('train_h128_' + model_type.replace('-', '_'), **out)
        print(model_type, 'hidden 128: loss', float(loss), 'step losses', step_losses, 'gradients', len(names))

    run('np-nd-np')
    _with_b5_shim(lambda: run('p-nd-np'))


def gen_generators():
    """The reference's CNF generators (src/pdp/generator.py) under fixed numpy seeds: uniform, modular and variable-modular,
    generate() and generate_complete() (the variable-modular generate_complete cannot run in the reference, App. B-11)."""
    import pdp.generator as RG
    out = {}
    specs = [('uniform', lambda: RG.UniformCNFGenerator(20, 40, 2, 5, 2.0, 5.0, 5)),
             ('modular', lambda: RG.ModularCNFGenerator(3, 30, 60, 0.3, 0.9, 3, 8, 2.0, 5.0, 5)),
             ('vmodular', lambda: RG.VariableModularCNFGenerator(2, 5, 30, 60, 0.3, 0.9, 3, 8, 2.0, 5.0, 5))]
    for name, make in specs:
        for method in ('generate', 'generate_complete'):
            if name == 'vmodular' and method == 'generate_complete':
                continue
            for seed in (1, 2, 3):
                np.random.seed(seed)
                g = make()
                for draw in range(2):
                    n, m, gm, ef, _, label, clauses = getattr(g, method)()
                    key = '%s_%s_s%d_d%d' % (name, method, seed, draw)
                    out[key + '_nm'] = np.array([n, m], dtype=np.int64)
                    out[key + '_gm'] = np.asarray(gm, dtype=np.int32)
                    out[key + '_ef'] = np.asarray(ef, dtype=np.float32)
    save('generators', **out)



def gen_subsumption():
    "dimacs2json -s (clause subsumption, dimacs2json.py:60-83) on files with duplicate, subsumed and opposite-sign clauses"
    import dimacs2json as RDJ
    ddir = os.path.join(HERE, 'dimacs_subsume')
    os.makedirs(ddir, exist_ok=True)
    rng = np.random.RandomState(77)
    for i in range(4):
        n = 12 + 2 * i
        clauses = gen.uniform_ksat(n, 30, 3, rng)
        extra = []
        for c in clauses[:10]:
            extra.append(c[:2])                                   # subsumes its parent (later in the file)
            extra.append(c + [int(((abs(c[0]) % n) + 1) * (1 if i % 2 else -1))] if abs(c[0]) % n + 1 not in [abs(x) for x in c] else c)
            extra.append([-c[0]] + c[1:])                         # opposite sign: no subsumption
        clauses = clauses[:5] + extra[:12] + clauses[5:] + extra[12:] + [clauses[3], clauses[7][:1]]
        gen.write_dimacs(os.path.join(ddir, 'sub_%d_%d.cnf' % (i, i % 2)), n, clauses)
    RDJ.convert_directory(ddir, os.path.join(HERE, 'dimacs_subsume.converted.jsonl'), True)
    RDJ.convert_directory(ddir, os.path.join(HERE, 'dimacs_subsume.plain.jsonl'), False)


# ---- E. CLI -------------------------------------------------------------------------------------

def gen_meta_data():
    """np-nd-np with graph features (meta_feature_dim = 3): no shipped config has them and the loader never yields any, but the plug-in API
    takes them (pdp_propagate.py:59-61, pdp_decimate.py:63-65, pdp_predict.py:57-59, solver.py:76-77).  A predict-style forward with
    batch replication 2 and Walk-SAT (recorded random stream), and a training-style forward + backward of the energy loss."""
    lines = make_lines([(16, 50, (3,))] * 4 + [(12, 34, (2, 3, 4))] * 2, seed0=3100)
    gm, bvm, bfm, ef, lab, misc = batch_tensors(lines)
    H, M, T, w = 16, 3, 4, 5
    cfg = base_cfg('np-nd-np', hidden_dim=H, meta_feature_dim=M, mem_hidden_dim=20, agg_hidden_dim=20, mem_agg_hidden_dim=10, classifier_dim=10,
                   local_search_iteration=w)
    tr, m = build(cfg, seed=4321)
    B = int(bvm.max()) + 1
    torch.manual_seed(5)
    meta = torch.randn(B, M)
    out = problem_arrays(gm, bvm, bfm, ef)
    out['meta_data'] = np_(meta)
    for k, v in m.state_dict().items():
        if '_module_list' not in k:                      # the aliases share storage with these
            out['w::' + k] = np_(v)
    RAND_LOG.clear()
    torch.rand = _rec_rand
    try:
        torch.manual_seed(9)
        with torch.no_grad():
            st = m.get_init_state(gm, bvm, bfm, ef, meta, randomized=True, batch_replication=2)        # (the test redraws it: same seed, same stream)
            out['init_checksum'] = np.array([float(x.double().sum()) for x in st[0] + st[1]])
            pred, (ps, ds) = m(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=meta,
                               is_training=False, iteration_num=T, check_termination=tr._check_recurrence_termination, batch_replication=2)
    finally:
        torch.rand = _TORCH_RAND
    out['rand_sizes'] = np.array([r.size for r in RAND_LOG], np.int64)
    out['final_prediction'] = np_(pred[0])[:, 0]
    for i, x in enumerate(ps):
        out['final_prop_%d' % i] = np_(x)
    for i, x in enumerate(ds):
        out['final_dec_%d' % i] = np_(x)
    # training style: two outer recurrences from zeros, loss = energy, gradients of the layers that see the meta columns
    m.zero_grad()
    st = m.get_init_state(gm, bvm, bfm, ef, meta, randomized=False)
    loss = torch.zeros(1)
    for t in range(2):
        pred, st = m(init_state=st, graph_map=gm, batch_variable_map=bvm, batch_function_map=bfm, edge_feature=ef, meta_data=meta,
                     is_training=True, iteration_num=2)
        loss = loss + tr._compute_loss(model=m, loss=None, prediction=pred, label=lab, graph_map=gm, batch_variable_map=bvm,
                                       batch_function_map=bfm, edge_feature=ef, meta_data=meta)
    loss.backward()
    out['train_loss'] = np_(loss)
    out['train_prediction'] = np_(pred[0])[:, 0]
    for name in ('_propagator._variable_aggregator._W1_m.weight', '_propagator._function_aggregator._W1_m.weight',
                 '_decimator._variable_rnn_cell.weight_ih', '_decimator._function_rnn_cell.weight_ih',
                 '_predictor._variable_aggregator._W1_m.weight', '_predictor._variable_classifier._layer2.weight'):
        obj = m
        for part in name.split('.'):
            obj = getattr(obj, part)
        out['grad::' + name] = np_(obj.grad)
    save('meta_data_np_nd_np', meta=np.array([T, w, H, M], np.int64), **out)


def gen_foreign_plugin():
    """tests/golden/foreign_plugin.py -- a plug-in triple written against the reference's API only -- run INSIDE the reference (CPU): records
    the batch, the parameters the reference's constructors drew, the per-sweep integer trajectory (which variable each instance fixed, the
    active flags, the active mask), the scores behind the decisions, the final states and the prediction.  The generator refuses a fixture
    that sits on a near tie (relative gap between the best and the second-best candidate of a decision below 1e-3)."""
    spec = importlib.util.spec_from_file_location('foreign_plugin', os.path.join(HERE, 'foreign_plugin.py'))
    fp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fp)                 # `from pdp.nn import solver, util` resolves to the reference here
    assert fp.util is RU and fp.solver is RS
    lines = make_lines([(30, 96, (3,)), (24, 70, (2, 3, 4)), (36, 120, (3,)), (28, 80, (3, 4)), (20, 64, (3,)), (32, 100, (2, 3))], seed0=7700)
    gm, bvm, bfm, ef, lab, misc = batch_tensors(lines)
    dev = torch.device('cpu')
    T, R, w = 14, 2, 6
    for seed in range(99, 140):                 # the first run seed whose decisions are all clear of a tie
        model = fp.build_solver(dev, local_search_iterations=w)
        RAND_LOG.clear()
        torch.rand = _rec_rand
        try:
            out = fp.run(model, dev, gm, bvm, bfm, ef, iterations=T, batch_replication=R, seed=seed)
        finally:
            torch.rand = _TORCH_RAND
        trace = out['trace']
        margins = np.array([t['margin'] for t in trace], np.float64)
        decided = np.stack([np_(t['decided']) for t in trace])
        if margins.min() > 1e-3 and (decided >= 0).sum() >= 8:
            break
    assert (decided >= 0).sum() >= 8, "the fixture should contain decimation steps"
    assert margins.min() > 1e-3, "a decision of the fixture sits on a near tie: %r" % (margins,)
    arrs = problem_arrays(gm, bvm, bfm, ef)
    for k, v in model.state_dict().items():
        arrs['w::' + k] = np_(v)
    for i, x in enumerate(out['states'][0]):
        arrs['final_prop_%d' % i] = np_(x)
    for i, x in enumerate(out['states'][1]):
        arrs['final_dec_%d' % i] = np_(x)
    save('foreign_plugin', meta=np.array([T, w, seed, R], np.int64), rand_sizes=np.array([r.size for r in RAND_LOG], np.int64),
         trace_decided=decided, trace_margin=margins,
         trace_active_variables=np.stack([np_(t['active_variables']) for t in trace]),
         trace_active_functions=np.stack([np_(t['active_functions']) for t in trace]),
         trace_active_mask=np.stack([np_(t['active_mask']) for t in trace]),
         trace_score=np.stack([np_(t['score']) for t in trace]), trace_pressure=np.stack([np_(t['pressure']) for t in trace]),
         check_log=np.array(out['check_log'], np.float64), final_prediction=np_(out['prediction'])[:, 0],
         final_solved=np_(out['solved'])[:, 0], final_unsat=np_(out['unsat'])[:, 0],
         counts_variables=np_(out['counts'][0])[:, 0], counts_functions=np_(out['counts'][1])[:, 0], **arrs)
    print('foreign_plugin: %d sweeps recorded, %d decisions, smallest decision gap %.3g, solved %s'
          % (len(trace), int((decided >= 0).sum()), margins.min(), np_(out['solved'])[:, 0].astype(int).tolist()))


def gen_cli():
    import runpy
    import shutil
    ddir = os.path.join(HERE, 'dimacs20')
    if os.path.isdir(ddir):
        shutil.rmtree(ddir)
    insts = []
    for i in range(20):
        n = 30 + (i % 3) * 10
        rng = np.random.RandomState(7000 + i)
        insts.append((n, gen.uniform_ksat(n, int(round(3.6 * n)), 3, rng)))
    os.makedirs(ddir)
    for i, (n, cl) in enumerate(insts):
        gen.write_dimacs(os.path.join(ddir, 'uf_%02d_%d.cnf' % (i, i % 2)), n, cl)
    out_path = os.path.join(HERE, 'cli_pdp_dimacs20.out.jsonl')
    argv = ['satyr.py', '/root/reference/config/Predict/PDP-p-d-p-sp-pytorch.yaml', ddir, '50', '-d', '-c',
            '-z', '8', '-s', '7', '-w', '40', '-o', out_path]
    old_argv, old_cwd = sys.argv, os.getcwd()
    sys.argv = argv
    sys.path.insert(0, REF)
    try:
        runpy.run_path(os.path.join(REF, 'satyr.py'), run_name='__main__')
    finally:
        sys.argv = old_argv
        os.chdir(old_cwd)
    # also keep the intermediate JSON the reference converter produced (loader golden vector)
    import dimacs2json as RDJ
    RDJ.convert_directory(ddir, os.path.join(HERE, 'cli_dimacs20.converted.jsonl'), False)
    print('wrote', out_path)


if __name__ == '__main__':
    what = sys.argv[1:] or ['problem', 'ops', 'traces', 'neural', 'npdnp', 'cli']
    lines = None
    if 'problem' in what or 'ops' in what:
        lines = gen_loader_and_simplify()
    if 'ops' in what:
        gen_ops(lines)
    if 'predictor_function' in what:
        gen_predictor_function_branch(make_lines(MIXED_SPECS, seed0=100))
    if 'scorer_adaptors' in what:
        gen_scorer_adaptors(make_lines(MIXED_SPECS, seed0=100))
    if 'traces' in what:
        gen_traces()
    if 'neural' in what:
        gen_neural()
    if 'npdnp' in what:
        gen_np_d_np()
    if 'pndnp' in what:
        gen_p_nd_np()
    if 'metrics' in what:
        gen_test_metrics()
    if 'headline' in what:
        gen_headline_poison()
    if 'headline_mid' in what:
        gen_headline_mid()
    if 'generators' in what:
        gen_generators()
    if 'subsume' in what:
        gen_subsumption()
    if 'headline_neural' in what:
        gen_headline_neural()
    if 'randinit' in what:
        run_trace('p-d-p', make_lines([(30, 108, (3,))] * 10 + [(24, 60, (2, 3, 4))] * 6, seed0=4300), T=25, w=15, seed=21, randomized=True,
                  tag='trace_pdp_randinit', cfg_kw=dict(tolerance=0.05, t_max=10))
    if 'rep_randinit' in what:
        # batch replication with a RANDOM initial state: the replicas differ and couple through the termination rule (trainer.py:157-160)
        run_trace('p-d-p', make_lines([(40, 140, (3,))] * 6, seed0=900), T=30, w=10, seed=29, replication=3, randomized=True,
                  tag='trace_pdp_rep3_randinit', cfg_kw=dict(tolerance=0.05, t_max=6))
    if 'rf_leak' in what:
        # Reinforce on four instances and a coin sequence found by tools/parity_soak.py (rf_leak_instances.jsonl, rf_leak_coins.npy; the
        # reference's torch.rand(1) calls are fed from that sequence): instance 0 leaves through the gate at sweep 75 while the others go on,
        # and the next sweep of its frozen state is not finite -- the reference's mask blend (mask * new + (1 - mask) * old,
        # pdp_propagate.py:219-221) turns messages of the INACTIVE instance into NaN.  Also pins torch.sign(NaN) = 0 for the force column.
        lines = [l for l in open(os.path.join(HERE, 'rf_leak_instances.jsonl')).read().split('\n') if l.strip()]
        coins = np.load(os.path.join(HERE, 'rf_leak_coins.npy'))
        pos = [0]
        keep = globals()['_real_rand']

        def fed(*a, **k):
            out = keep(*a, **k)
            out.fill_(float(coins[pos[0] % len(coins)])); pos[0] += 1
            return out
        globals()['_real_rand'] = fed
        try:
            run_trace('reinforce', lines, T=100, w=0, seed=0, tag='trace_reinforce_nan_leak', cfg_kw=dict(pi=0.1, decimation_probability=0.6),
                      float_iters=(0, 1, 74, 75, 76, 77, 80))
        finally:
            globals()['_real_rand'] = keep
        d = np.load(os.path.join(HERE, 'trace_reinforce_nan_leak.npz'))
        am = d['trace_active_mask']
        print('rf_leak: iterations', int(d['iterations_run'][0]), 'instance 0 inactive from', int(np.argmax(am[:, 0] == 0)), 'NaN entries in the final state',
              int(np.isnan(d['final_dec_1']).sum()), int(np.isnan(d['final_dec_0']).sum()), 'at sweep 75/76/77:', [int(np.isnan(d['prop_fs_%d' % t]).sum()) for t in (75, 76, 77)])
    if 'foreign' in what:
        gen_foreign_plugin()
    if 'meta' in what:
        gen_meta_data()
    if 'cli' in what:
        gen_cli()
    if 'config0' in what:
        gen_config0_cli()
    if 'config4' in what:
        gen_config4_mixed()
    if 'neural_long' in what:
        gen_neural_long()
    if 'train' in what:
        gen_train()
    if 'train_hybrid' in what:
        gen_train_hybrid()
    if 'train_h128' in what:
        gen_train_h128()

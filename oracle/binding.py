"""ctypes binding of the CPU oracle (oracle/libpdp_oracle.so).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from the
product package (pdp-solver_amd/).  Arrays are numpy, fp32 / int32 / uint8, C-contiguous.
"""

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libpdp_oracle.so')

_lib = None


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(os.path.join(HERE, f)) > os.path.getmtime(LIB_PATH)
            for f in ('pdp_oracle.c', 'pdp_oracle_neural.c', '../include/pdp_math.h')):
        subprocess.check_call(['make', '-s', '-C', HERE])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_problem_create.restype = C.c_void_p
        _lib.orc_decimator_create.restype = C.c_void_p
        _lib.orc_refresh_edge_mask.restype = C.c_double
        _lib.orc_sequential_decimate.restype = C.c_int
        _lib.orc_local_search.restype = C.c_int
    return _lib


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class ForwardArgs(C.Structure):
    _fields_ = [
        ('model', C.c_int), ('iterations', C.c_int), ('local_search_iterations', C.c_int),
        ('epsilon', C.c_float), ('tolerance', C.c_float), ('t_max', C.c_float), ('pi', C.c_float),
        ('decimation_probability', C.c_float),
        ('rng_mode', C.c_int), ('stream', C.c_void_p), ('n_stream', C.c_int64), ('seed', C.c_uint64),
        ('prediction', C.c_void_p), ('q', C.c_void_p), ('fs', C.c_void_p),
        ('iterations_run', C.c_void_p), ('rand_consumed', C.c_void_p), ('walksat_steps', C.c_void_p),
        ('trace_active_var', C.c_void_p), ('trace_active_fn', C.c_void_p), ('trace_solution', C.c_void_p),
        ('trace_active_mask', C.c_void_p), ('trace_q', C.c_void_p), ('trace_fs', C.c_void_p),
    ]


MODELS = {'p-d-p': 0, 'walk-sat': 1, 'reinforce': 2}


class Problem(object):
    """A batch of CNF instances (the oracle's SATProblem)."""

    def __init__(self, graph_map, batch_variable_map, batch_function_map, edge_feature, replication=1):
        L = lib()
        gm = np.ascontiguousarray(graph_map, dtype=np.int32)
        bvm = np.ascontiguousarray(batch_variable_map, dtype=np.int32)
        bfm = np.ascontiguousarray(batch_function_map, dtype=np.int32)
        ef = _f(np.asarray(edge_feature).reshape(-1))
        self._h = C.c_void_p(L.orc_problem_create(C.c_int(gm.shape[1]), C.c_int(bvm.size), C.c_int(bfm.size),
                                                 _p(gm), _p(bvm), _p(bfm), _p(ef), C.c_int(replication)))
        dims = np.zeros(5, dtype=np.int32)
        L.orc_problem_dims(self._h, _p(dims))
        self.E, self.V, self.F, self.B, self.R = [int(x) for x in dims]

    def set_rng_base(self, first_variable, first_instance):
        lib().orc_problem_set_rng_base(self._h, C.c_uint32(int(first_variable)), C.c_uint32(int(first_instance)))

    def __del__(self):
        try:
            if self._h:
                lib().orc_problem_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- state ---------------------------------------------------------------------------
    def state(self):
        av = np.zeros(self.V, np.float32); af = np.zeros(self.F, np.float32)
        sol = np.zeros(self.V, np.float32); sat = np.zeros(self.B, np.float32)
        lib().orc_problem_get_state(self._h, _p(av), _p(af), _p(sol), _p(sat))
        return av, af, sol, sat

    def set_state(self, active_var=None, active_fn=None, solution=None):
        lib().orc_problem_set_state(self._h, _p(_f(active_var)), _p(_f(active_fn)), _p(_f(solution)))

    def graph(self):
        ev = np.zeros(self.E, np.int32); ec = np.zeros(self.E, np.int32); es = np.zeros(self.E, np.float32)
        vi = np.zeros(self.V, np.int32); fi = np.zeros(self.F, np.int32)
        lib().orc_problem_get_graph(self._h, _p(ev), _p(ec), _p(es), _p(vi), _p(fi))
        return ev, ec, es, vi, fi

    # -- K7 -----------------------------------------------------------------------------
    def simplify(self):
        lib().orc_simplify(self._h)

    def set_variables(self, assignment):
        a = _f(assignment).copy()
        lib().orc_set_variables(self._h, _p(a))
        return a

    # -- K8 -----------------------------------------------------------------------------
    def refresh_edge_mask(self):
        s = lib().orc_refresh_edge_mask(self._h)
        m = np.zeros(self.E, np.float32)
        lib().orc_get_edge_mask(self._h, _p(m))
        return m, s

    def set_edge_mask(self, m):
        lib().orc_set_edge_mask(self._h, _p(_f(m)))

    # -- K4/K5 --------------------------------------------------------------------------
    def smooth_max(self, x):
        out = np.zeros(self.V, np.float32)
        lib().orc_smooth_max(self._h, _p(_f(x)), _p(out))
        return out

    def instance_max(self, x):
        out = np.zeros(self.B, np.float32)
        lib().orc_instance_max(self._h, _p(_f(x)), _p(out))
        return out

    def instance_argmax(self, x):
        out = np.zeros(self.B, np.int64)
        lib().orc_instance_argmax(self._h, _p(_f(x)), _p(out))
        return out

    # -- K1-K3, K6 ----------------------------------------------------------------------
    def sp_propagate(self, dec_q, dec_fs, edge_mask, active_mask, init_q, init_fs, pi=0.0):
        oq = np.zeros((self.E, 3), np.float32); ofs = np.zeros((self.E, 2), np.float32)
        am = None if active_mask is None else np.ascontiguousarray(active_mask, dtype=np.uint8)
        lib().orc_sp_propagate(self._h, _p(_f(dec_q)), _p(_f(dec_fs)), _p(_f(edge_mask)), _p(am),
                               _p(_f(init_q)), _p(_f(init_fs)), C.c_float(pi), _p(oq), _p(ofs))
        return oq, ofs

    def sp_propagate_adapted(self, xlog, dec_fs, edge_mask, active_mask, init_q, init_fs, pi=0.0):
        oq = np.zeros((self.E, 3), np.float32); ofs = np.zeros((self.E, 2), np.float32)
        am = None if active_mask is None else np.ascontiguousarray(active_mask, dtype=np.uint8)
        lib().orc_sp_propagate_adapted(self._h, _p(_f(xlog)), _p(_f(dec_fs)), _p(_f(edge_mask)), _p(am),
                                       _p(_f(init_q)), _p(_f(init_fs)), C.c_float(pi), _p(oq), _p(ofs))
        return oq, ofs

    def survey_score(self, fs, pi=0.0):
        out = np.zeros(self.V, np.float32)
        lib().orc_survey_score(self._h, _p(_f(fs)), C.c_float(pi), _p(out))
        return out

    # -- K9, K13 ------------------------------------------------------------------------
    def cnf_eval(self, pred):
        s = np.zeros(self.B, np.float32); u = np.zeros(self.B, np.float32)
        lib().orc_cnf_eval(self._h, _p(_f(pred)), _p(s), _p(u))
        return s, u

    def sat_loss(self, pred, coeff, eps, sharpness):
        lib().orc_sat_loss.restype = C.c_float
        return float(lib().orc_sat_loss(self._h, _p(_f(pred)), C.c_float(coeff), C.c_float(eps), C.c_int(int(sharpness))))

    def update_solution(self, pred):
        out = np.zeros(self.V, np.float32)
        lib().orc_update_solution(self._h, _p(_f(pred)), _p(out))
        return out

    def check_termination(self, active_mask, pred):
        am = np.ascontiguousarray(active_mask, dtype=np.uint8).copy()
        lib().orc_check_termination(self._h, _p(am), _p(_f(pred)))
        return am

    # -- decimators ---------------------------------------------------------------------
    def new_decimator(self):
        return C.c_void_p(lib().orc_decimator_create(self._h))

    def free_decimator(self, d):
        lib().orc_decimator_destroy(d)

    def decimator_set(self, d, prev, counters, has_prev):
        lib().orc_decimator_set(self._h, d, _p(_f(prev)), _p(_f(counters)), C.c_int(int(has_prev)))

    def decimator_get(self, d):
        prev = np.zeros(self.E, np.float32); cnt = np.zeros(self.B, np.float32); fl = np.zeros(2, np.int32)
        lib().orc_decimator_get(self._h, d, _p(prev), _p(cnt), _p(fl))
        return prev, cnt, fl

    def sequential_decimate(self, d, fs, active_mask, tolerance, t_max, pi=0.0):
        am = None if active_mask is None else np.ascontiguousarray(active_mask, dtype=np.uint8).copy()
        n = lib().orc_sequential_decimate(self._h, d, _p(_f(fs)), _p(am), C.c_float(tolerance), C.c_float(t_max), C.c_float(pi))
        return am, n

    # -- K14 ----------------------------------------------------------------------------
    def energy(self, assignment):
        en = np.zeros(self.B, np.float32); uf = np.zeros(self.F, np.float32)
        lib().orc_energy(self._h, _p(_f(assignment)), _p(en), _p(uf))
        return en, uf

    def energy_diff(self, assignment):
        d = np.zeros(self.V, np.float32)
        lib().orc_energy_diff(self._h, _p(_f(assignment)), _p(d))
        return d

    def local_search(self, pred, iterations, epsilon, stream=None, seed=0, cursor=0):
        out = np.zeros(self.V, np.float32)
        cur = C.c_int64(cursor)
        st = _f(stream)
        n = lib().orc_local_search(self._h, _p(_f(pred)), C.c_int(iterations), C.c_float(epsilon),
                                   C.c_int(0 if stream is not None else 1), _p(st),
                                   C.c_int64(0 if st is None else st.size), C.byref(cur), C.c_uint64(seed), _p(out))
        return out, n, cur.value

    def random_fill(self, stream=None, seed=0, cursor=0):
        cur = C.c_int64(cursor)
        st = _f(stream)
        lib().orc_random_fill(self._h, C.c_int(0 if stream is not None else 1), _p(st),
                              C.c_int64(0 if st is None else st.size), C.byref(cur), C.c_uint64(seed))
        return cur.value

    def deduplicate(self, pred):
        out = np.zeros(self.V // self.R, np.float32); ch = np.zeros(self.B // self.R, np.int32)
        lib().orc_deduplicate(self._h, _p(_f(pred)), _p(out), _p(ch))
        return out, ch

    # -- whole forward ------------------------------------------------------------------
    def forward(self, model, iterations, local_search_iterations=0, epsilon=0.5, tolerance=0.02, t_max=100,
                pi=0.0, decimation_probability=0.5, stream=None, seed=0, trace=False, trace_float=False):
        a = ForwardArgs()
        a.model = MODELS[model]; a.iterations = iterations; a.local_search_iterations = local_search_iterations
        a.epsilon = epsilon; a.tolerance = tolerance; a.t_max = t_max; a.pi = pi
        a.decimation_probability = decimation_probability
        st = _f(stream)
        a.rng_mode = 0 if stream is not None else 1
        a.stream = _p(st); a.n_stream = 0 if st is None else st.size; a.seed = seed
        T = max(iterations, 1)
        res = dict(prediction=np.zeros(self.V // self.R, np.float32), q=np.zeros((self.E, 3), np.float32),
                   fs=np.zeros((self.E, 2), np.float32), iterations_run=np.zeros(1, np.int32),
                   rand_consumed=np.zeros(1, np.int64), walksat_steps=np.zeros(1, np.int32))
        for k in ('prediction', 'q', 'fs', 'iterations_run', 'rand_consumed', 'walksat_steps'):
            setattr(a, k, _p(res[k]))
        if trace == 'mask':                  # the per-sweep instance mask only (T*B bytes: affordable at full size)
            res['trace_active_mask'] = np.zeros((T, self.B), np.uint8)
            a.trace_active_mask = _p(res['trace_active_mask'])
        elif trace:
            res['trace_active_var'] = np.zeros((T, self.V), np.float32)
            res['trace_active_fn'] = np.zeros((T, self.F), np.float32)
            res['trace_solution'] = np.zeros((T, self.V), np.float32)
            res['trace_active_mask'] = np.zeros((T, self.B), np.uint8)
            for k in ('trace_active_var', 'trace_active_fn', 'trace_solution', 'trace_active_mask'):
                setattr(a, k, _p(res[k]))
        if trace_float:
            res['trace_q'] = np.zeros((T, self.E, 3), np.float32)
            res['trace_fs'] = np.zeros((T, self.E, 2), np.float32)
            a.trace_q = _p(res['trace_q']); a.trace_fs = _p(res['trace_fs'])
        lib().orc_forward(self._h, C.byref(a))
        res['iterations_run'] = int(res['iterations_run'][0])
        res['rand_consumed'] = int(res['rand_consumed'][0])
        res['walksat_steps'] = int(res['walksat_steps'][0])
        return res


def math_apply(fn, x):
    names = {'exp': 0, 'log': 1, 'logsigmoid': 2, 'sigmoid': 3, 'tanh': 4, 'safe_exp': 5, 'safe_log': 6,
             'philox': 7, 'rcp': 8, 'safe_exp_fast': 9, 'safe_log_fin': 10, 'safe_log_fin_scorer': 11, 'exp_fin': 12, 'tanh_abs': 13, 'rcp_ge1': 14}
    x = _f(x)
    y = np.zeros_like(x)
    lib().orc_math_apply(C.c_int(names[fn]), _p(x), _p(y), C.c_int64(x.size))
    return y


# ---- neural operators (oracle/pdp_oracle_neural.c) -----------------------------------------------------------------------
class AggWeights(C.Structure):
    _fields_ = [('din', C.c_int), ('m1', C.c_int), ('a', C.c_int), ('fd', C.c_int), ('g', C.c_int), ('out', C.c_int),
                ('W1m', C.c_void_p), ('b1m', C.c_void_p), ('W2m', C.c_void_p), ('W1a', C.c_void_p), ('b1a', C.c_void_p),
                ('W2a', C.c_void_p)]


ACTS = {'none': 0, 'logsigmoid': 1, 'relu': 2, 'sigmoid': 3, 'tanh': 4}


def linear(x, W, b=None, act='none'):
    x = _f(x); W = _f(W); b = _f(b)
    R, K = x.shape
    y = np.zeros((R, W.shape[0]), np.float32)
    lib().orc_linear(_p(x), C.c_int64(R), C.c_int(K), C.c_int64(K), _p(W), _p(b), C.c_int(W.shape[0]), C.c_int(ACTS[act]), _p(y), C.c_int64(W.shape[0]))
    return y


def csr_rows(row_of_edge, n_rows):
    "ascending-edge-id CSR of a row index vector (helper for the neural aggregator)"
    row_of_edge = np.asarray(row_of_edge, dtype=np.int64)
    order = np.argsort(row_of_edge, kind='stable').astype(np.int32)
    ptr = np.zeros(n_rows + 1, np.int32)
    np.cumsum(np.bincount(row_of_edge, minlength=n_rows), out=ptr[1:])
    return ptr, order


def aggregator(edge_row, n_rows, state, edge_sign, edge_mask, include_self, w):
    """w: dict with W1m,b1m,W2m,W1a,b1a,W2a (numpy).  Returns [E,out] (or [n_rows,out] when include_self)."""
    state = _f(state); edge_sign = _f(edge_sign); em = _f(edge_mask)
    E = state.shape[0]
    ptr, order = csr_rows(edge_row, n_rows)
    er = np.ascontiguousarray(edge_row, dtype=np.int32)
    keep = [_f(w[k]) for k in ('W1m', 'b1m', 'W2m', 'W1a', 'b1a', 'W2a')]
    aw = AggWeights()
    aw.din = keep[0].shape[1]; aw.m1 = keep[0].shape[0]; aw.a = keep[2].shape[0]; aw.g = keep[3].shape[0]; aw.out = keep[5].shape[0]
    aw.fd = keep[3].shape[1] - aw.a
    for name, arr in zip(('W1m', 'b1m', 'W2m', 'W1a', 'b1a', 'W2a'), keep):
        setattr(aw, name, arr.ctypes.data)
    out = np.zeros((n_rows if include_self else E, aw.out), np.float32)
    lib().orc_aggregator(C.c_int(E), C.c_int(n_rows), _p(ptr), _p(order), _p(er), _p(state), _p(edge_sign), _p(em),
                         C.c_int(1 if include_self else 0), C.byref(aw), _p(out))
    return out


def gru(state, edge_sign, h, W_ih, W_hh, b_ih, b_hh, mask=None):
    state = _f(state); h = _f(h)
    E, H = h.shape
    out = np.zeros((E, H), np.float32)
    lib().orc_gru(C.c_int(E), C.c_int(H), C.c_int(state.shape[1]), _p(state), _p(_f(edge_sign)), _p(h), _p(_f(W_ih)), _p(_f(W_hh)),
                  _p(_f(b_ih)), _p(_f(b_hh)), _p(_f(mask)), _p(out))
    return out


def perceptron(x, W1, b1, W2, out_act='sigmoid'):
    x = _f(x)
    y = np.zeros((x.shape[0], 1), np.float32)
    lib().orc_perceptron(C.c_int(x.shape[0]), C.c_int(x.shape[1]), C.c_int(_f(W1).shape[0]), _p(x), _p(_f(W1)), _p(_f(b1)), _p(_f(W2)),
                         C.c_int(ACTS[out_act]), _p(y))
    return y


def neural_weights(d, prefix='w__'):
    "golden .npz (flattened state-dict keys) -> nested dict of the np-nd-np model's tensors"
    g = lambda k: d[prefix + k]
    def agg(base):
        return dict(W1m=g(base + '___W1_m__weight'), b1m=g(base + '___W1_m__bias'), W2m=g(base + '___W2_m__weight'),
                    W1a=g(base + '___W1_a__weight'), b1a=g(base + '___W1_a__bias'), W2a=g(base + '___W2_a__weight'))
    def cell(base):
        return dict(W_ih=g(base + '__weight_ih'), W_hh=g(base + '__weight_hh'), b_ih=g(base + '__bias_ih'), b_hh=g(base + '__bias_hh'))
    return dict(prop_v=agg('_propagator___variable_aggregator'), prop_f=agg('_propagator___function_aggregator'),
                gru_v=cell('_decimator___variable_rnn_cell'), gru_f=cell('_decimator___function_rnn_cell'),
                pred=agg('_predictor___variable_aggregator'),
                head=dict(W1=g('_predictor___variable_classifier___layer1__weight'), b1=g('_predictor___variable_classifier___layer1__bias'),
                          W2=g('_predictor___variable_classifier___layer2__weight')))


def neural_forward(problem, weights, init, T, trace=None):
    """np-nd-np forward loop (solver.py:355-386 with the neural plug-ins) composed from the oracle's operators.
    problem: binding.Problem (already simplified); init = (prop_v, prop_f, dec_v, dec_f) numpy [E,H].
    Returns (final prediction [V] after _local_search(0 steps) + _update_solution, states dict)."""
    ev, ec, es, vi, fi = problem.graph()
    E, V, F, B = problem.E, problem.V, problem.F, problem.B
    pv, pf, dv, df = [np.ascontiguousarray(x, dtype=np.float32) for x in init]
    am = np.ones(B, np.uint8)
    em = None
    iters = 0
    pred = None
    for t in range(T):
        mask = am[vi[ev]].astype(np.float32)
        fstate = aggregator(ev, V, dv, es, em, False, weights['prop_v'])
        pf = (mask[:, None] * fstate + (1.0 - mask[:, None]) * pf).astype(np.float32)
        vstate = aggregator(ec, F, df, es, em, False, weights['prop_f'])
        pv = (mask[:, None] * vstate + (1.0 - mask[:, None]) * pv).astype(np.float32)
        dv = gru(pv, es, dv, mask=mask, **weights['gru_v'])
        df = gru(pf, es, df, mask=mask, **weights['gru_f'])
        m, s = problem.refresh_edge_mask()
        if s < E:
            em = m
        agg = aggregator(ev, V, dv, es, em, True, weights['pred'])
        p = perceptron(agg, weights['head']['W1'], weights['head']['b1'], weights['head']['W2'])[:, 0]
        pred = problem.update_solution(p)
        if trace is not None:
            trace.append(dict(prop_v=pv.copy(), prop_f=pf.copy(), dec_v=dv.copy(), dec_f=df.copy(), pred=pred.copy()))
        am = problem.check_termination(am, pred)
        if trace is not None:
            trace[-1]['active_mask'] = am.copy()
        iters = t + 1
        if am.sum() <= 0:
            break
    # final predictor call + local search with 0 steps + update_solution (solver.py:342-348)
    agg = aggregator(ev, V, dv, es, em, True, weights['pred'])
    p = perceptron(agg, weights['head']['W1'], weights['head']['b1'], weights['head']['W2'])[:, 0]
    ls, _, _ = problem.local_search(p, 0, 0.5, seed=0)
    final = problem.update_solution(ls)
    return final, dict(prop_v=pv, prop_f=pf, dec_v=dv, dec_f=df, iterations=iters, active_mask=am)


def pnd_weights(d, prefix='w__'):
    "golden .npz -> tensors of the p-nd-np model (adaptor projections, two GRU cells, predictor)"
    g = lambda k: d[prefix + k]
    nw = dict(w_f=g('_propagator___function_input_projector__weight'), W_v=g('_propagator___variable_input_projector__weight'))
    def agg(base):
        return dict(W1m=g(base + '___W1_m__weight'), b1m=g(base + '___W1_m__bias'), W2m=g(base + '___W2_m__weight'),
                    W1a=g(base + '___W1_a__weight'), b1a=g(base + '___W1_a__bias'), W2a=g(base + '___W2_a__weight'))
    def cell(base):
        return dict(W_ih=g(base + '__weight_ih'), W_hh=g(base + '__weight_hh'), b_ih=g(base + '__bias_ih'), b_hh=g(base + '__bias_hh'))
    nw.update(gru_v=cell('_decimator___variable_rnn_cell'), gru_f=cell('_decimator___function_rnn_cell'),
              pred=agg('_predictor___variable_aggregator'),
              head=dict(W1=g('_predictor___variable_classifier___layer1__weight'), b1=g('_predictor___variable_classifier___layer1__bias'),
                        W2=g('_predictor___variable_classifier___layer2__weight')))
    return nw


def sp_adaptors(dec_v, dec_f, w_f, W_v):
    """include_adaptors=True inputs of the SP propagator (pdp_propagate.py:166-167, 179-182): k-ascending fmaf dot products,
    xlog = logsigmoid(w_f . dec_v), eta = sigmoid(W_v[0] . dec_f), force = sign(W_v[1] . dec_f)"""
    xlog = linear(dec_v, np.reshape(w_f, (1, -1)), None, 'logsigmoid')[:, 0]
    b = linear(dec_f, W_v, None, 'none')
    eta = math_apply('sigmoid', b[:, 0])
    force = np.sign(b[:, 1]).astype(np.float32)           # NaN stays NaN, like torch.sign
    return np.ascontiguousarray(xlog, dtype=np.float32), np.ascontiguousarray(np.stack((eta, force), 1), dtype=np.float32)


def pnd_forward(problem, weights, init, T, trace=None, walksat=None):
    """p-nd-np forward loop (solver.py:355-386: SP propagator with adaptors, neural decimator, neural predictor).
    init = (q [E,3], fs [E,2], dec_v [E,H], dec_f [E,H]).  walksat = dict(steps, epsilon, stream, cursor): the post-processing
    local search of solver.py:433-467 on the recorded random stream (cursor is updated in place); with batch replication the
    result is de-duplicated (solver.py:401-431)."""
    ev, ec, es, vi, fi = problem.graph()
    E, V, B = problem.E, problem.V, problem.B
    q, fs, dv, df = [np.ascontiguousarray(x, dtype=np.float32) for x in init]
    am = np.ones(B, np.uint8)
    em = None
    iters = 0
    for t in range(T):
        mask = am[vi[ev]].astype(np.float32)
        xlog, fs2 = sp_adaptors(dv, df, weights['w_f'], weights['W_v'])
        q, fs = problem.sp_propagate_adapted(xlog, fs2, em, am, q, fs, 0.0)
        dv = gru(q, es, dv, mask=mask, **weights['gru_v'])
        df = gru(fs, es, df, mask=mask, **weights['gru_f'])
        m, s = problem.refresh_edge_mask()
        if s < E:
            em = m
        agg = aggregator(ev, V, dv, es, em, True, weights['pred'])
        p = perceptron(agg, weights['head']['W1'], weights['head']['b1'], weights['head']['W2'])[:, 0]
        pred = problem.update_solution(p)
        if trace is not None:
            trace.append(dict(prop_q=q.copy(), prop_fs=fs.copy(), dec_v=dv.copy(), dec_f=df.copy(), pred=pred.copy()))
        am = problem.check_termination(am, pred)
        if trace is not None:
            trace[-1]['active_mask'] = am.copy()
        iters = t + 1
        if am.sum() <= 0:
            break
    agg = aggregator(ev, V, dv, es, em, True, weights['pred'])
    p = perceptron(agg, weights['head']['W1'], weights['head']['b1'], weights['head']['W2'])[:, 0]
    steps = 0
    if walksat is None:
        ls, _, _ = problem.local_search(p, 0, 0.5, seed=0)
    else:
        ls, steps, walksat['cursor'] = problem.local_search(p, walksat['steps'], walksat['epsilon'], stream=walksat['stream'], cursor=walksat['cursor'])
    final = problem.update_solution(ls)
    if problem.R > 1:
        final, _ = problem.deduplicate(final)
    return final, dict(q=q, fs=fs, dec_v=dv, dec_f=df, iterations=iters, active_mask=am, walksat_steps=steps)

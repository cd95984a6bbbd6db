"""PyTorch-CPU restatement of the reference's own formulation of the p-d-p hot path: sparse COO incidence masks and torch.mm
products, including the dense [V x B] matrices of util.sparse_max / sparse_argmax.  TEST INFRASTRUCTURE ONLY (bench.py's second
`cpu_baseline` entry and tests/): this is "what the reference's --cpu_mode computes, op for op" (SURVEY.md section 8(d)(2)), written
from the algorithm, timed with torch.set_num_threads(n_cores) as the reference does (src/pdp/factorgraph/base.py:43-50).

Restates (reference file:line):
  masks ............ SATProblem._compute_*_mask            src/pdp/nn/solver.py:84-178
  simplify ......... _propagate_single_clauses / _peel     src/pdp/nn/solver.py:180-285
  sweep ............ SurveyPropagator.forward              src/pdp/nn/pdp_propagate.py:139-221
  decimate ......... SequentialDecimator.forward           src/pdp/nn/pdp_decimate.py:122-177
  score ............ SurveyScorer.forward                  src/pdp/nn/pdp_predict.py:155-192
  smooth / max ..... util.sparse_smooth_max / sparse_max / sparse_argmax   src/pdp/nn/util.py:257-286
  check ............ SatCNFEvaluator.forward, _check_recurrence_termination   src/pdp/nn/util.py:210-236, src/pdp/trainer.py:150-162
  loop ............. _forward_core                         src/pdp/nn/solver.py:355-386
Pinned by tests/test_oracle_golden.py::test_torch_sparse_port_equals_oracle (integer trajectory equal to the C oracle's, messages
within the tolerance that separates torch's exp/log from include/pdp_math.h)."""

import numpy as np
import torch


def _coo(rows, cols, vals, shape):
    return torch.sparse_coo_tensor(torch.stack([rows, cols]), vals, shape).coalesce()


class SparseBatch(object):
    "incidence masks of a collated batch (edges x variables, edges x clauses, nodes x instances) as sparse COO matrices"

    def __init__(self, graph_map, batch_variable_map, batch_function_map, edge_feature):
        gm = torch.as_tensor(np.asarray(graph_map)).long()
        self.var_of, self.fn_of = gm[0], gm[1]
        self.inst_of_var = torch.as_tensor(np.asarray(batch_variable_map)).long()
        self.inst_of_fn = torch.as_tensor(np.asarray(batch_function_map)).long()
        self.sign = torch.as_tensor(np.asarray(edge_feature, dtype=np.float32)).reshape(-1, 1)
        E, V, F = self.var_of.numel(), self.inst_of_var.numel(), self.inst_of_fn.numel()
        B = int(self.inst_of_var.max().item()) + 1
        self.E, self.V, self.F, self.B = E, V, F, B
        e = torch.arange(E)
        one = torch.ones(E)
        s = self.sign[:, 0]
        self.VE = _coo(self.var_of, e, one, (V, E)); self.EV = self.VE.t().coalesce()          # variable x edge and its transpose
        self.FE = _coo(self.fn_of, e, one, (F, E)); self.EF = self.FE.t().coalesce()
        self.VE_pos = _coo(self.var_of, e, (s == 1).float(), (V, E))
        self.VE_neg = _coo(self.var_of, e, (s == -1).float(), (V, E))
        self.EV_signed = _coo(e, self.var_of, s, (E, V))
        self.VF = _coo(self.var_of, self.fn_of, one, (V, F)); self.FV = self.VF.t().coalesce()
        self.VF_signed = _coo(self.var_of, self.fn_of, s, (V, F)); self.FV_signed = self.VF_signed.t().coalesce()
        self.VB = _coo(torch.arange(V), self.inst_of_var, torch.ones(V), (V, B)); self.BV = self.VB.t().coalesce()
        self.FB = _coo(torch.arange(F), self.inst_of_fn, torch.ones(F), (F, B)); self.BF = self.FB.t().coalesce()
        self.active_var = torch.ones(V, 1); self.active_fn = torch.ones(F, 1)
        self.solution = 0.5 * torch.ones(V); self.is_sat = 0.5 * torch.ones(B)
        self.edge_mask = None

    # ---- simplification (unit propagation, satisfied clauses, pure literals) --------------------------------------------------
    def _fix(self, assignment):
        assignment = assignment * self.active_var
        assigned = assignment[:, 0].abs() == 1
        n_in = torch.mm(self.FV, assignment.abs())
        value = torch.mm(self.FV_signed, assignment)
        satisfied = (value > -n_in).float() * self.active_fn
        self.active_var[assigned, 0] = 0
        self.active_fn[satisfied[:, 0] == 1, 0] = 0
        self.solution[assigned] = (assignment[assigned, 0] + 1) / 2.0

    def _units(self):
        while True:
            unit = (torch.mm(self.FV, self.active_var) == 1).float() * self.active_fn
            if unit.sum() <= 0:
                return
            n_in = torch.mm(self.VF, unit)
            value = torch.mm(self.VF_signed, unit)
            clash = (value.abs() != n_in).float() * self.active_var
            if clash.sum() > 0:
                dead = torch.mm(self.BV, clash)
                self.is_sat[dead[:, 0] >= 1] = 0
                self.active_fn[(torch.mm(self.FB, dead) * self.active_fn)[:, 0] == 1, 0] = 0
                self.active_var[(torch.mm(self.VB, dead) * self.active_var)[:, 0] == 1, 0] = 0
            forced = (value.abs() == n_in).float() * self.active_var
            self.active_fn[unit[:, 0] == 1, 0] = 0
            self._fix(torch.sign(value) * forced)

    def _pure(self):
        deg = torch.mm(self.VF, self.active_fn)
        sdeg = torch.mm(self.VF_signed, self.active_fn)
        while True:
            pure = (deg == sdeg.abs()).float() * self.active_var
            if pure.sum() <= 0:
                return
            gone = (torch.mm(self.FV, pure) > 0).float() * self.active_fn
            d_deg = torch.mm(self.VF, gone) * self.active_var
            d_sdeg = torch.mm(self.VF_signed, gone) * self.active_var
            sel = pure[:, 0] == 1
            self.solution[sel] = (sdeg[sel, 0].sign() + 1) / 2.0
            deg -= d_deg; sdeg -= d_sdeg
            self.active_var[sel, 0] = 0
            self.active_fn[gone[:, 0] == 1, 0] = 0

    def simplify(self):
        self._units(); self._pure()

    def set_variables(self, assignment):
        self._fix(assignment); self.simplify()


_EPS_SP = torch.tensor([1e-40]); _EPS_SC = torch.tensor([1e-10]); _CAP = torch.tensor([30.0]); _ONE = torch.ones(1)


def _slog(x, eps):
    return torch.max(x, eps).log()


def _sexp(x):
    return torch.min(x, _CAP).exp()


def _dense_cols(P, x):
    "the [V x B] matrix of util.sparse_max / sparse_argmax: entry (v, instance of v) = x_v - min(x) + 1, zeros elsewhere"
    return torch.sparse_coo_tensor(P.VB.indices(), x - x.min() + 1, P.VB.shape).to_dense()


def instance_max(P, x):
    return torch.max(_dense_cols(P, x), 0)[0] + x.min() - 1


def instance_argmax(P, x):
    return torch.argmax(_dense_cols(P, x), 0)


def smooth_max(P, x):
    w = _sexp(30 * x)
    return torch.mm(P.VE, x * w) / torch.max(torch.mm(P.VE, w), _ONE)


def sp_sweep(P, q, fs, prev_q, prev_fs, edge_mask, active_mask, pi=0.0):
    "one SurveyPropagator.forward: (q, fs) = previous propagator state, (prev_q, prev_fs) = decimator state"
    pi_t = torch.tensor([pi])
    mask = torch.mm(P.EV, torch.mm(P.VB, active_mask.float())) if active_mask is not None else torch.ones(P.E, 1)
    x = _slog(prev_q[:, 0], _EPS_SP).unsqueeze(1)
    if edge_mask is not None:
        x = x * edge_mask
    agg = torch.mm(P.EF, torch.mm(P.FE, x)) - x
    eta = mask * _sexp(agg) + (1 - mask) * fs[:, 0].unsqueeze(1)
    force = prev_fs[:, 1].unsqueeze(1)
    y = _slog(1 - prev_fs[:, 0], _EPS_SP).unsqueeze(1)
    if edge_mask is not None:
        y = y * edge_mask
    pos = torch.mm(P.EV, torch.mm(P.VE_pos, y))
    neg = torch.mm(P.EV, torch.mm(P.VE_neg, y))
    s = P.sign
    same = 0.5 * (1 + s) * pos + 0.5 * (1 - s) * neg
    same = same - y
    same = same + _slog(1.0 - pi_t * (force == s).float(), _EPS_SP)
    opp = 0.5 * (1 - s) * pos + 0.5 * (1 + s) * neg
    opp = opp + _slog(1.0 - pi_t * (force == -s).float(), _EPS_SP)
    dc = _sexp(same + opp)
    a, b = _sexp(same), _sexp(opp)
    qu, qs = a * (1 - b), b * (1 - a)
    out = torch.cat((qu, qs, dc), 1) / (qu + qs + dc)
    return mask * out + (1 - mask) * q, torch.cat((eta, force), 1)


def survey_score(P, fs, pi=0.0):
    pi_t = torch.tensor([pi])
    ext = torch.sign(torch.mm(P.VE, fs[:, 1].unsqueeze(1)))
    y = _slog(1 - fs[:, 0], _EPS_SC).unsqueeze(1) * torch.mm(P.EF, P.active_fn)
    pos = torch.mm(P.VE_pos, y) + _slog(1.0 - pi_t * (ext == 1).float(), _EPS_SC)
    neg = torch.mm(P.VE_neg, y) + _slog(1.0 - pi_t * (ext == -1).float(), _EPS_SC)
    both = pos + neg
    dc = torch.mm(P.VE, y) + _slog(1.0 - pi_t, _EPS_SC)
    bias = (2 * both + dc) / 4.0
    pos, neg, both = pos - bias, neg - bias, both - bias
    dc = _sexp(dc - bias)
    q0 = _sexp(pos) - _sexp(both)
    q1 = _sexp(neg) - _sexp(both)
    tot = _slog(q0 + q1 + dc, _EPS_SC)
    return _sexp(_slog(q1, _EPS_SC) - tot) - _sexp(_slog(q0, _EPS_SC) - tot)


class Decimator(object):
    def __init__(self, tolerance, t_max):
        self.tol, self.t_max, self.prev, self.cnt = tolerance, t_max, None, None

    def step(self, P, fs, active_mask):
        if self.cnt is None:
            self.cnt = torch.zeros(P.B, 1)
        eta = fs[:, 0]
        g = smooth_max(P, eta.unsqueeze(1)) * P.active_var
        g = instance_max(P, g.squeeze(1)).unsqueeze(1)
        active_mask[g <= 1e-10] = 0
        if self.prev is not None and P.active_var.sum() > 0:
            d = (self.prev - eta).abs().unsqueeze(1)
            if P.edge_mask is not None:
                d = d * P.edge_mask
            d = smooth_max(P, d) * P.active_var
            d = instance_max(P, d.squeeze(1)).unsqueeze(1)
            self.cnt[d[:, 0] < self.tol, 0] = 0
            conv = (d < self.tol).float()
            conv[self.cnt[:, 0] >= self.t_max, 0] = 1
            self.cnt[self.cnt[:, 0] >= self.t_max, 0] = 0
            conv_v = torch.mm(P.VB, conv)
            if conv_v.sum() > 0:
                score = survey_score(P, fs)
                coeff = score.abs() * P.active_var * conv_v
                if coeff.sum() > 0:
                    pick = instance_argmax(P, coeff.squeeze(1))
                    norm = torch.mm(P.BV, coeff)
                    pick = pick[(active_mask * (norm != 0)).squeeze(1).bool()]
                    if pick.numel() > 0:
                        assign = torch.zeros(P.V, 1)
                        assign[pick, 0] = score.sign()[pick, 0]
                        P.set_variables(assign)
            self.cnt = self.cnt + 1
        self.prev = eta


def cnf_solved(P, pred):
    v = torch.mm(P.EV_signed, pred) + (1 - P.sign) / 2
    sat = (torch.mm(P.FE, (v > 0.5).float()) > 0).float()
    total = torch.mm(P.BF, torch.ones(P.F, 1))
    got = torch.mm(P.BF, sat)
    return (total == got).float(), total - got


def forward_loop(P, T, tolerance=0.02, t_max=100, simplify=True, trace=None, times=None, max_seconds=None):
    """p-d-p: simplify + T iterations of propagate / decimate / refresh / predict / terminate from the deterministic initial state.
    `times` (a list) receives the wall seconds of every iteration (the first one has no previous survey: no convergence test);
    with `max_seconds` the loop stops once the iterations run so far took longer than that (bounded CPU baseline)."""
    import time
    if simplify:
        P.simplify()
    q = torch.ones(P.E, 3) / 3.0
    fs = 0.5 * torch.ones(P.E, 2); fs[:, 1] = 0
    dq, dfs = q, fs
    active = torch.ones(P.B, 1, dtype=torch.uint8)
    dec = Decimator(tolerance, t_max)
    em = None
    done = 0
    for _ in range(T):
        t0 = time.perf_counter()
        q, fs = sp_sweep(P, q, fs, dq, dfs, em, active)
        dec.step(P, fs, active)
        dq, dfs = q, fs
        P.edge_mask = torch.mm(P.EV, P.active_var) * torch.mm(P.EF, P.active_fn)
        if P.edge_mask.sum() < P.E:
            em = P.edge_mask
        pred = P.active_var * P.solution.unsqueeze(1) + (1.0 - P.active_var) * P.solution.unsqueeze(1)
        solved, _ = cnf_solved(P, pred)
        live = active[:, 0].clone().bool()
        active[live, 0] = (solved[live, 0] <= 0.5).to(active.dtype)
        done += 1
        if times is not None:
            times.append(time.perf_counter() - t0)
        if trace is not None:
            trace.append(dict(active_var=P.active_var[:, 0].clone().numpy(), active_fn=P.active_fn[:, 0].clone().numpy(),
                              solution=P.solution.clone().numpy(), active_mask=active[:, 0].clone().numpy(), q=q.clone().numpy()))
        if active.sum() <= 0:
            break
        if max_seconds is not None and times is not None and sum(times) > max_seconds:
            break
    return q, fs, done

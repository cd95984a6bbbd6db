"""A third-party propagator / decimator / predictor triple written ONLY against the reference's plug-in API surface
(microsoft/PDP-Solver, src/pdp/nn/{solver,util}.py): it imports ``pdp.nn.util`` and ``pdp.nn.solver`` by name and touches nothing else.

The same file runs in two worlds:
  * tests/golden/generate_golden.py imports it with the REFERENCE's ``pdp`` package on sys.path (CPU) and records its trajectory
    -> tests/golden/foreign_plugin.npz;
  * tests/test_foreign_plugin.py imports it with this repo's ``pdp`` package on sys.path and runs it on the GPU through
    ``PropagatorDecimatorSolverBase._forward_core_stepwise``; integer outputs must be equal, floats within the stated tolerance.

What it exercises (the surface SURVEY.md section 8b lists): ``sat_problem._graph_mask_tuple / _batch_mask_tuple / _vf_mask_tuple /
_signed_mask_tuple / _pos_mask_tuple / _neg_mask_tuple / _replication_mask_tuple`` (with torch.mm and handed to util functions),
``util.MessageAggregator(...)(state, feature, mask, mask_transpose, edge_mask)`` in its full and its partial (no pre-transform) form and
with include_self_message on and off, on SATProblem's masks and on masks built with ``util.SatLossEvaluator.compute_masks``,
``util.sparse_smooth_max`` (default alpha and alpha=5), ``util.sparse_max`` / ``util.sparse_argmax`` on the variable and on the clause
instance mask, ``util.safe_exp``, ``util.SatLossEvaluator.compute_batch_mask / safe_log``, ``util.SatCNFEvaluator`` called with bare
tensors, ``sat_problem.set_variables``, in-place updates of ``active_mask``, 2- and 3-tuple states, and ``PropagatorDecimatorSolverBase``
with batch replication, the termination callback and the Walk-SAT post-process.

This is test material written for this repository; it is not part of the product and not derived from reference source.
"""

import torch
import torch.nn as nn

from pdp.nn import solver, util


class ForeignPropagator(nn.Module):
    "messages both ways through deep-set aggregators; the clause side uses masks the plug-in builds itself"

    def __init__(self, device, hidden, mem_hidden, mem_agg_hidden, agg_hidden):
        super(ForeignPropagator, self).__init__()
        self._device = device
        self._hidden = hidden
        # full four-layer form on the variable side, the partial form (raw states are summed, then two layers) on the clause side
        self._variable_aggregator = util.MessageAggregator(device, hidden + 1, hidden, mem_hidden, mem_agg_hidden, agg_hidden, 1,
                                                           include_self_message=False)
        self._function_aggregator = util.MessageAggregator(device, hidden + 1, hidden, 0, mem_agg_hidden, agg_hidden, 1,
                                                           include_self_message=False)
        self._own_masks = None

    def forward(self, init_state, decimator_state, sat_problem, is_training, active_mask=None):
        variable_mask, variable_mask_transpose, function_mask, function_mask_transpose = sat_problem._graph_mask_tuple
        b_variable_mask = sat_problem._batch_mask_tuple[0]
        vf_mask = sat_problem._vf_mask_tuple[0]

        if active_mask is not None:
            mask = torch.mm(variable_mask_transpose, torch.mm(b_variable_mask, active_mask.float()))
        else:
            mask = torch.ones(init_state[0].size(0), 1, device=self._device)

        if len(decimator_state) == 3:
            dec_v, dec_f, edge_mask = decimator_state
        else:
            dec_v, dec_f = decimator_state
            edge_mask = None
        variable_state, function_state = init_state
        sign = sat_problem._edge_feature

        # damping by the number of clauses still active around the edge's variable
        degree = torch.mm(variable_mask_transpose, torch.mm(vf_mask, sat_problem._active_functions))
        damp = 1.0 / (1.0 + 0.25 * degree)

        # variables --> functions, on the problem's own masks
        new_f = self._variable_aggregator(torch.cat((dec_v, sign), 1), sign, variable_mask, variable_mask_transpose, edge_mask)
        function_state = mask * (damp * new_f) + (1 - mask) * function_state

        # functions --> variables, on masks built from the raw tensors (what SatLossEvaluator.compute_masks returns)
        if self._own_masks is None or self._own_masks[0] is not sat_problem:
            _, own_function_mask = util.SatLossEvaluator.compute_masks(sat_problem._graph_map, sat_problem._batch_variable_map,
                                                                       sat_problem._batch_function_map, sat_problem._edge_feature, self._device)
            self._own_masks = (sat_problem, own_function_mask, own_function_mask.transpose(0, 1))
        new_v = self._function_aggregator(torch.cat((dec_f, sign), 1), sign, self._own_masks[1], self._own_masks[2], edge_mask)
        variable_state = mask * new_v + (1 - mask) * variable_state
        return variable_state, function_state

    def get_init_state(self, graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, randomized, batch_replication):
        edge_num = graph_map.size(1) * batch_replication
        # drawn on the host so that both worlds see the same numbers for the same seed
        v = (torch.rand(edge_num, self._hidden, dtype=torch.float32) - 0.5).to(self._device)
        f = (torch.rand(edge_num, self._hidden, dtype=torch.float32) - 0.5).to(self._device)
        return (v, f)


class ForeignDecimator(nn.Module):
    "a GRU-free state update plus a periodic greedy decimation step driven by util.sparse_* on the reference's masks"

    def __init__(self, device, hidden, period):
        super(ForeignDecimator, self).__init__()
        self._device = device
        self._hidden = hidden
        self._period = period
        self._mix = nn.Linear(hidden, hidden, bias=False)
        self._calls = 0
        self.trace = []

    def forward(self, init_state, message_state, sat_problem, is_training, active_mask=None):
        variable_state, function_state = message_state[0], message_state[1]
        new_v = 0.5 * init_state[0] + 0.5 * torch.tanh(self._mix(variable_state))
        new_f = 0.5 * init_state[1] + 0.5 * torch.tanh(self._mix(function_state))
        self._calls += 1
        record = dict(decided=-torch.ones(sat_problem._batch_size, dtype=torch.int64), margin=float('inf'))

        variable_mask = sat_problem._graph_mask_tuple[0]
        b_variable_mask, b_variable_mask_transpose, b_function_mask, _ = sat_problem._batch_mask_tuple

        # per-variable evidence from the clause messages: a sharp and a soft smooth-max over the variable's edges
        evidence = new_f[:, 0].unsqueeze(1)
        if sat_problem._edge_mask is not None:
            evidence = evidence * sat_problem._edge_mask
        sharp = util.sparse_smooth_max(evidence, variable_mask, self._device)
        soft = util.sparse_smooth_max(evidence, variable_mask, self._device, alpha=5)
        polarity = torch.mm(sat_problem._pos_mask_tuple[0], evidence) - torch.mm(sat_problem._neg_mask_tuple[0], evidence)
        score = (0.7 * sharp + 0.3 * soft + 0.05 * torch.tanh(polarity)) * sat_problem._active_variables

        # instances whose evidence is flat are switched off (like the sequential decimator's gate, on the variable instance mask)
        if active_mask is not None:
            spread = util.sparse_max(score.abs().squeeze(1), b_variable_mask, self._device).unsqueeze(1)
            active_mask[spread <= 1e-12] = 0

        # per-instance clause pressure through the CLAUSE instance mask
        clause_load = torch.mm(sat_problem._graph_mask_tuple[2], util.safe_exp(evidence, self._device)) * sat_problem._active_functions
        pressure = util.sparse_max(clause_load.squeeze(1), b_function_mask, self._device)
        record['pressure'] = pressure.detach().cpu().clone()

        if self._calls % self._period == 0 and sat_problem._active_variables.sum() > 0:
            coeff = score.abs()
            max_ind = util.sparse_argmax(coeff.squeeze(1), b_variable_mask, self._device)
            norm = torch.mm(b_variable_mask_transpose, coeff)
            # how decisive the arg-max is (recorded by the generator to make sure the fixture does not sit on a tie)
            top = util.sparse_max(coeff.squeeze(1), b_variable_mask, self._device)
            masked = coeff.squeeze(1).clone()
            masked[max_ind] = 0
            second = util.sparse_max(masked, b_variable_mask, self._device)
            live = norm.squeeze(1) != 0
            if active_mask is not None:
                live = live & (active_mask.squeeze(1) != 0)
            if live.any():
                gap = ((top - second) / top.clamp(min=1e-30))[live]
                record['margin'] = float(gap.min().item())
                chosen = max_ind[live]
                assignment = torch.zeros(sat_problem._variable_num, 1, device=self._device)
                assignment[chosen, 0] = score.sign()[chosen, 0]
                sat_problem.set_variables(assignment)
                decided = -torch.ones(sat_problem._batch_size, dtype=torch.int64)
                decided[live.cpu()] = chosen.cpu()
                record['decided'] = decided

        record['active_variables'] = sat_problem._active_variables[:, 0].detach().cpu().clone()
        record['active_functions'] = sat_problem._active_functions[:, 0].detach().cpu().clone()
        record['active_mask'] = None if active_mask is None else active_mask[:, 0].detach().cpu().clone()
        record['score'] = score[:, 0].detach().cpu().clone()
        self.trace.append(record)
        return new_v, new_f

    def get_init_state(self, graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, randomized, batch_replication):
        self._calls = 0
        self.trace = []
        edge_num = graph_map.size(1) * batch_replication
        v = (torch.rand(edge_num, self._hidden, dtype=torch.float32) - 0.5).to(self._device)
        f = (torch.rand(edge_num, self._hidden, dtype=torch.float32) - 0.5).to(self._device)
        return (v, f)


class ForeignPredictor(nn.Module):
    "variable beliefs from a deep-set aggregate with the self message plus a signed-degree prior"

    def __init__(self, device, hidden, mem_hidden, mem_agg_hidden, agg_hidden):
        super(ForeignPredictor, self).__init__()
        self._device = device
        self._aggregator = util.MessageAggregator(device, hidden + 1, hidden, mem_hidden, mem_agg_hidden, agg_hidden, 0,
                                                  include_self_message=True)
        self._head = nn.Linear(hidden, 1, bias=True)

    def forward(self, decimator_state, sat_problem, last_call=False):
        variable_mask, variable_mask_transpose, _, _ = sat_problem._graph_mask_tuple
        if len(decimator_state) == 3:
            dec_v, _, edge_mask = decimator_state
        else:
            dec_v, _ = decimator_state
            edge_mask = None
        agg = self._aggregator(torch.cat((dec_v, sat_problem._edge_feature), 1), None, variable_mask, variable_mask_transpose, edge_mask)
        live_edges = sat_problem._edge_mask if sat_problem._edge_mask is not None else torch.ones(sat_problem._edge_num, 1, device=self._device)
        prior = torch.mm(sat_problem._signed_mask_tuple[0], live_edges)
        return torch.sigmoid(self._head(agg) + 0.3 * prior), None


def make_check_termination(device):
    """A termination callback in the reference's shape (src/pdp/trainer.py:150-162): the evaluator is called with the problem's bare tensors
    and the replication masks go through torch.mm.  (The boolean index is cloned: the reference's self-aliased index_put no longer
    runs on a current torch)"""
    evaluator = util.SatCNFEvaluator(device=device)
    eps = 1e-6 * torch.ones(1, device=device)

    def check(active, prediction, sat_problem):
        output, unsat = evaluator(variable_prediction=prediction[0], graph_map=sat_problem._graph_map,
                                  batch_variable_map=sat_problem._batch_variable_map, batch_function_map=sat_problem._batch_function_map,
                                  edge_feature=sat_problem._edge_feature, meta_data=sat_problem._meta_data)
        check.log_unsat.append(util.SatLossEvaluator.safe_log(unsat + 1.0, eps).sum().item())
        idx = active[:, 0].clone().bool()
        if sat_problem._batch_replication > 1:
            real_batch = torch.mm(sat_problem._replication_mask_tuple[1], (output > 0.5).float())
            dup_batch = torch.mm(sat_problem._replication_mask_tuple[0], (real_batch == 0).float())
            active[idx, 0] = (dup_batch[idx, 0] > 0).to(active.dtype)
        else:
            active[idx, 0] = (output[idx, 0] <= 0.5).to(active.dtype)
    check.log_unsat = []
    return check


def build_solver(device, hidden=8, mem_hidden=12, mem_agg_hidden=6, agg_hidden=10, period=3, local_search_iterations=0, epsilon=0.5, seed=321):
    "the triple inside the reference's solver base class; parameters come from the host generator under `seed`"
    torch.manual_seed(seed)
    propagator = ForeignPropagator(device, hidden, mem_hidden, mem_agg_hidden, agg_hidden)
    decimator = ForeignDecimator(device, hidden, period)
    predictor = ForeignPredictor(device, hidden, mem_hidden, mem_agg_hidden, agg_hidden)
    model = solver.PropagatorDecimatorSolverBase(device, 'foreign-triple', propagator, decimator, predictor,
                                                 local_search_iterations=local_search_iterations, epsilon=epsilon)
    return model.to(device)


def run(model, device, graph_map, batch_variable_map, batch_function_map, edge_feature, iterations, batch_replication, seed):
    "one forward the way FactorGraphTrainerBase._predict_batch drives a solver (src/pdp/factorgraph/base.py:280-305)"
    check = make_check_termination(device)
    torch.manual_seed(seed)
    with torch.no_grad():
        state = model.get_init_state(graph_map, batch_variable_map, batch_function_map, edge_feature, None, randomized=True,
                                     batch_replication=batch_replication)
        prediction, states = model(init_state=state, graph_map=graph_map, batch_variable_map=batch_variable_map,
                                   batch_function_map=batch_function_map, edge_feature=edge_feature, meta_data=None, is_training=False,
                                   iteration_num=iterations, check_termination=check, batch_replication=batch_replication)
        compact = util.SatLossEvaluator.compute_batch_mask(batch_variable_map, batch_function_map, device)
        # ones per instance through the compact masks: variables and clauses of every original instance
        counts = (torch.mm(compact[1], torch.ones(batch_variable_map.size(0), 1, device=device)),
                  torch.mm(compact[3], torch.ones(batch_function_map.size(0), 1, device=device)))
        solved, unsat = util.SatCNFEvaluator(device=device)(prediction[0], graph_map, batch_variable_map, batch_function_map, edge_feature, None)
    return dict(prediction=prediction[0], states=states, trace=model._decimator.trace, check_log=check.log_unsat, counts=counts,
                solved=solved, unsat=unsat)

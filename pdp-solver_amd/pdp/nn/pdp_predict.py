"""Predictors and scorers of the PDP framework (reference: src/pdp/nn/pdp_predict.py)."""

import torch
import torch.nn as nn

from pdp import native


def _init_sp_state(device, edge_num, randomized):
    "reference: SurveyScorer.get_init_state pdp_predict.py:194-208"
    if randomized:
        variable_state = torch.rand(edge_num, 3, dtype=torch.float32)
        function_state = torch.rand(edge_num, 2, dtype=torch.float32)
        function_state[:, 1] = 0
        return (variable_state.to(device), function_state.to(device))
    variable_state = torch.ones(edge_num, 3, dtype=torch.float32, device=device) / 3.0
    function_state = 0.5 * torch.ones(edge_num, 2, dtype=torch.float32, device=device)
    function_state[:, 1] = 0
    return (variable_state, function_state)


class NeuralPredictor(nn.Module):
    """Aggregates the edge states at the variable (and / or function) nodes (deep set with self message) and classifies them
    (reference: pdp_predict.py:18-104).  The variable branch -- the one every solver of the reference's factory builds -- runs as the fused
    native predictor; the function branch (function_classifier is None in solver.py:534,558) runs on the generic layer / row-sum operators of
    pdp/nn/train_ops.py."""

    def __init__(self, device, decimator_dimension, prediction_dimension, edge_dimension, meta_data_dimension, mem_hidden_dimension,
                 agg_hidden_dimension, mem_agg_hidden_dimension, variable_classifier=None, function_classifier=None):
        super(NeuralPredictor, self).__init__()
        if edge_dimension != 1 or prediction_dimension != 1:
            raise native.NativeError("NeuralPredictor: edge_feature_dim = 1 and prediction_dim = 1 only")
        self._meta_dim = meta_data_dimension       # > 0: graph features appended to the aggregator's input (pdp_predict.py:57-59, 71-72) -> generic operators
        from pdp.nn import util
        self._device = device
        self._module_list = nn.ModuleList()
        self._variable_classifier = variable_classifier
        self._function_classifier = function_classifier
        self._hidden_dimension = decimator_dimension
        make = lambda: util.MessageAggregator(device, decimator_dimension + edge_dimension + meta_data_dimension, decimator_dimension,
                                              mem_hidden_dimension, mem_agg_hidden_dimension, agg_hidden_dimension, 0, include_self_message=True)
        if variable_classifier is not None:
            self._variable_aggregator = make()
            self._module_list.append(self._variable_aggregator)
            self._module_list.append(self._variable_classifier)
        if function_classifier is not None:
            self._function_aggregator = make()
            self._module_list.append(self._function_aggregator)
            self._module_list.append(self._function_classifier)
        self._head = None
        self._head_key = None

    def _head_weights(self):
        c = self._variable_classifier
        params = (c._layer1.weight, c._layer1.bias, c._layer2.weight)
        key = tuple((p.data_ptr(), p._version) for p in params)
        if self._head is None or key != self._head_key:
            act = 'tanh' if type(c).__name__ == 'PerceptronTanh' else 'sigmoid'
            self._head = native.HeadWeights(*[p.data for p in params], out_act=act)
            self._head_key = key
        return self._head

    @staticmethod
    def _generic_branch(state, aggregator, classifier, sat_problem, by_variable, edge_mask):
        "aggregator + perceptron head as generic native operators (differentiable; pdp_predict.py:67-89, trainer.py:28-29 for the head)"
        from pdp.nn import train_ops as T
        gf = sat_problem.edge_meta()
        if gf is None:
            agg = aggregator.forward_train(state, None, sat_problem, by_variable, edge_mask, state_feature=sat_problem._edge_feature)
        else:
            agg = aggregator.forward_train(torch.cat((state, sat_problem._edge_feature, gf), 1), None, sat_problem, by_variable, edge_mask)
        hid = T.LinearAct.apply(agg, classifier._layer1.weight, classifier._layer1.bias, 'relu')
        out_act = 'tanh' if type(classifier).__name__ == 'PerceptronTanh' else 'sigmoid'
        return T.LinearAct.apply(hid, classifier._layer2.weight, None, out_act)

    def forward(self, decimator_state, sat_problem, last_call=False):
        pinned = getattr(self, '_train_path', None)       # set by the solver's forward (one decision for the three plug-ins)
        train = (torch.is_grad_enabled() and decimator_state[0].requires_grad) if pinned is None else pinned
        train = train or self._meta_dim > 0 or sat_problem._meta_data is not None       # graph features: generic operators
        em = decimator_state[2] if len(decimator_state) == 3 else None
        variable_prediction = function_prediction = None
        if self._variable_classifier is not None:
            if train:
                variable_prediction = self._generic_branch(decimator_state[0], self._variable_aggregator, self._variable_classifier, sat_problem, True, em)
            else:
                variable_prediction = sat_problem._native.neural_predict(self._variable_aggregator.native_weights(), self._head_weights(),
                                                                         decimator_state[0].contiguous(), None if em is None else em.reshape(-1).contiguous())
        if self._function_classifier is not None:
            function_prediction = self._generic_branch(decimator_state[1], self._function_aggregator, self._function_classifier, sat_problem, False, em)
        return variable_prediction, function_prediction

    def get_init_state(self, graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, randomized, batch_replication):
        edge_num = graph_map.size(1) * batch_replication
        if randomized:
            variable_state = 2.0 * torch.rand(edge_num, self._hidden_dimension, dtype=torch.float32) - 1.0
            function_state = 2.0 * torch.rand(edge_num, self._hidden_dimension, dtype=torch.float32) - 1.0
            return (variable_state.to(self._device), function_state.to(self._device))
        return (torch.zeros(edge_num, self._hidden_dimension, dtype=torch.float32, device=self._device),
                torch.zeros(edge_num, self._hidden_dimension, dtype=torch.float32, device=self._device))


class IdentityPredictor(nn.Module):
    """Prediction = the problem's current solution; on the last call the still-undecided variables are filled
    with uniform random numbers (reference: pdp_predict.py:110-128).  ``rng`` selects where those numbers come
    from: 'torch' draws them from the global torch CPU generator in the reference's order (bit-compatible
    with the reference's --cpu_mode run for the same seed), 'philox' draws them on the device."""

    def __init__(self, device, random_fill=False, rng='torch', seed=0):
        super(IdentityPredictor, self).__init__()
        self._random_fill = random_fill
        self._device = device
        self._rng = rng
        self._seed = seed

    def forward(self, decimator_state, sat_problem, last_call=False):
        pred = sat_problem._solution.unsqueeze(1)
        if self._random_fill and last_call:
            if self._rng == 'torch':
                active_var_num = int((sat_problem._active_variables[:, 0] > 0).long().sum().item())
                if active_var_num > 0:
                    sat_problem._native.random_fill(values=torch.rand(active_var_num).to(self._device))
            else:
                sat_problem._native.random_fill(seed=self._seed)
        return pred, None


class SurveyScorer(nn.Module):
    "SP bias W+ - W- per variable (reference: pdp_predict.py:134-208)."

    def __init__(self, device, message_dimension, include_adaptors=False, pi=0.0):
        super(SurveyScorer, self).__init__()
        self._device = device
        self._include_adaptors = bool(include_adaptors)
        self._pi = float(pi)
        if self._include_adaptors:
            # a bias-free projector [message_dimension -> 2] in front of the score (pdp_predict.py:145-147): column 0 through a sigmoid is the
            # survey, column 1 through sign the external force (:161-164).  No solver of the reference's factory asks for it.
            self._projector = nn.Linear(message_dimension, 2, bias=False).to(device)
            self._module_list = nn.ModuleList([self._projector])

    def forward(self, message_state, sat_problem, last_call=False):
        nat = sat_problem._native
        fs = message_state[1].contiguous()
        if self._include_adaptors:
            # the same two k-ascending dot products per edge as the propagator's adaptor form (k_sp_adaptors; its first output is not used here)
            W = self._projector.weight.data.contiguous()
            _, fs = nat.sp_adaptors(fs, fs, W[0].contiguous(), W)
        return nat.survey_score(fs, self._pi), None

    def get_init_state(self, graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, randomized, batch_replication):
        return _init_sp_state(self._device, graph_map.size(1) * batch_replication, randomized)


class ReinforcePredictor(nn.Module):
    "x_i = [sum of the external force over the variable's edges > 0] (reference: pdp_predict.py:214-226)."

    def __init__(self, device):
        super(ReinforcePredictor, self).__init__()
        self._device = device

    def forward(self, decimator_state, sat_problem, last_call=False):
        return sat_problem._native.reinforce_predict(decimator_state[1].contiguous()), None

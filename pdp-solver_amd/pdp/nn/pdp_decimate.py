"""Decimators of the PDP framework (reference: src/pdp/nn/pdp_decimate.py)."""

import torch
import torch.nn as nn

from pdp import native
from pdp.nn import pdp_predict, util


class NeuralDecimator(nn.Module):
    """Two GRU cells on the edges (reference: pdp_decimate.py:21-100); nn.GRUCell parameters keep their names so that
    reference checkpoints load, the cell itself runs on the matrix cores."""

    def __init__(self, device, message_dimension, meta_data_dimension, hidden_dimension, mem_hidden_dimension,
                 mem_agg_hidden_dimension, agg_hidden_dimension, edge_dimension, dropout):
        super(NeuralDecimator, self).__init__()
        if edge_dimension != 1:
            raise native.NativeError("NeuralDecimator: edge_feature_dim = 1 only (the loader's edge feature is the literal's sign)")
        self._meta_dim = meta_data_dimension       # > 0: graph features appended to the cells' inputs (pdp_decimate.py:63-65, 72-73, 80-81) -> generic GRU operator
        self._device = device
        self._module_list = nn.ModuleList()
        self._drop_out = dropout
        if isinstance(message_dimension, tuple):
            variable_message_dim, function_message_dim = message_dimension
        else:
            variable_message_dim = function_message_dim = message_dimension
        self._variable_rnn_cell = nn.GRUCell(variable_message_dim + edge_dimension + meta_data_dimension, hidden_dimension, bias=True)
        self._function_rnn_cell = nn.GRUCell(function_message_dim + edge_dimension + meta_data_dimension, hidden_dimension, bias=True)
        self._module_list.append(self._variable_rnn_cell)
        self._module_list.append(self._function_rnn_cell)
        self._hidden_dimension = hidden_dimension
        self._mem_hidden_dimension = mem_hidden_dimension
        self._agg_hidden_dimension = agg_hidden_dimension
        self._mem_agg_hidden_dimension = mem_agg_hidden_dimension
        self._native = {}

    def _weights(self, name, cell):
        params = (cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)
        key = tuple((p.data_ptr(), p._version) for p in params)
        if name not in self._native or self._native[name][0] != key:
            self._native[name] = (key, native.GruWeights(*[p.data for p in params]))
        return self._native[name][1]

    def forward(self, init_state, message_state, sat_problem, is_training, active_mask=None):
        variable_state, function_state = message_state
        if util.on_train_path(self, is_training) or self._meta_dim > 0 or sat_problem._meta_data is not None:
            # the differentiable cells of the training path (pdp_decimate.py:51-87); also what runs when graph features widen the inputs
            from pdp.nn import train_ops as T
            sign = sat_problem._edge_feature
            gf = sat_problem.edge_meta()
            extra = (sign,) if gf is None else (sign, gf)
            cv, cf = self._variable_rnn_cell, self._function_rnn_cell
            # hidden 128 without graph features: the cell runs on its two input pieces (train_ops.GruCellS: forward = one launch of the
            # pipelined inference kernel, no [E, 129] concatenation); PDP_TRAIN_CAT=1: the concatenated form of rounds 1-3
            import os
            ok = lambda t_: t_.size(1) == 128 or 2 <= t_.size(1) <= 3                # np-nd-np's [E, 128] messages, p-nd-np's surveys [E, 3] / [E, 2]
            fused = gf is None and self._hidden_dimension == 128 and ok(variable_state) and ok(function_state)
            if fused and os.environ.get('PDP_TRAIN_CAT', '0') != '1':
                nv = T.GruCellS.apply(variable_state, sign, init_state[0], cv.weight_ih, cv.weight_hh, cv.bias_ih, cv.bias_hh, self._weights('v', cv))
                nf = T.GruCellS.apply(function_state, sign, init_state[1], cf.weight_ih, cf.weight_hh, cf.bias_ih, cf.bias_hh, self._weights('f', cf))
            else:
                pv = (self._weights('v', cv), variable_state, sign) if fused else ()
                pf = (self._weights('f', cf), function_state, sign) if fused else ()
                nv = T.GruCell.apply(torch.cat((variable_state,) + extra, 1), init_state[0], cv.weight_ih, cv.weight_hh, cv.bias_ih, cv.bias_hh, *pv)
                nf = T.GruCell.apply(torch.cat((function_state,) + extra, 1), init_state[1], cf.weight_ih, cf.weight_hh, cf.bias_ih, cf.bias_hh, *pf)
            if active_mask is not None:
                mask = active_mask.reshape(-1).float()[sat_problem._batch_variable_map.long()][sat_problem._graph_map[0].long()].unsqueeze(1)
                nv = mask * nv + (1 - mask) * init_state[0]; nf = mask * nf + (1 - mask) * init_state[1]
            return nv, nf
        am = None if active_mask is None else active_mask.reshape(-1).contiguous()
        nat = sat_problem._native
        out_v, out_f = getattr(self, '_out', None) or (None, None)        # the solver's graph loop names the buffers a sweep writes
        new_variable_state = nat.neural_gru(self._weights('v', self._variable_rnn_cell), variable_state.contiguous(),
                                            init_state[0].contiguous(), am, out=out_v)
        new_function_state = nat.neural_gru(self._weights('f', self._function_rnn_cell), function_state.contiguous(),
                                            init_state[1].contiguous(), am, out=out_f)
        return new_variable_state, new_function_state

    def get_init_state(self, graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, randomized, batch_replication):
        "reference: pdp_decimate.py:89-100"
        edge_num = graph_map.size(1) * batch_replication
        if randomized:
            where = self._device if getattr(self, '_init_rng', 'torch') == 'device' else None       # (see NeuralMessagePasser.get_init_state)
            variable_state = 2.0 * torch.rand(edge_num, self._hidden_dimension, dtype=torch.float32, device=where) - 1.0
            function_state = 2.0 * torch.rand(edge_num, self._hidden_dimension, dtype=torch.float32, device=where) - 1.0
            return (variable_state.to(self._device), function_state.to(self._device))
        return (torch.zeros(edge_num, self._hidden_dimension, dtype=torch.float32, device=self._device),
                torch.zeros(edge_num, self._hidden_dimension, dtype=torch.float32, device=self._device))


class SequentialDecimator(nn.Module):
    """Convergence-gated greedy decimation, one variable per converged instance and call
    (reference: pdp_decimate.py:106-183).  The stateful parts of the reference (``_previous_function_state``,
    ``_counters``) live in a native decimator handle that is re-created by ``get_init_state``."""

    def __init__(self, device, message_dimension, scorer, tolerance, t_max):
        super(SequentialDecimator, self).__init__()
        self._device = device
        self._tolerance = tolerance
        self._scorer = scorer
        self._message_dimension = message_dimension
        self._t_max = t_max
        self._module_list = nn.ModuleList([self._scorer])
        self._handle = None
        self._handle_problem = None

    def native_handle(self, sat_problem):
        if self._handle is None or self._handle_problem is not sat_problem._native:
            self._handle = native.Decimator(sat_problem._native)
            self._handle_problem = sat_problem._native
        return self._handle

    def forward(self, init_state, message_state, sat_problem, is_training, active_mask=None):
        handle = self.native_handle(sat_problem)
        fs = message_state[1]
        if fs.size(1) != 2:
            # neural propagator (model type np-d-np): the gate reads column 0 of an [E, H] state (pdp_decimate.py:128,
            # 137); the native gate / apply kernels take the [E, 2] survey layout, column 1 is unused with a foreign scorer
            fs = torch.stack((fs[:, 0], torch.zeros_like(fs[:, 0])), dim=1)
        fs = fs.contiguous()
        am = None if active_mask is None else active_mask.reshape(-1)
        if isinstance(self._scorer, pdp_predict.SurveyScorer):
            sat_problem._native.sequential_decimate(handle, fs, am, self._tolerance, self._t_max, self._scorer._pi)
        else:
            # foreign scorer plug-in: native gate, python scorer, native apply (pdp_decimate.py:152-171)
            if sat_problem._native.sequential_decimate_gate(handle, fs, am, self._tolerance, self._t_max):
                score, _ = self._scorer(message_state, sat_problem)
                sat_problem._native.sequential_decimate_apply(handle, fs, score.reshape(-1).contiguous(), am)
            else:
                sat_problem._native.sequential_decimate_apply(handle, fs, None, am)
        return message_state

    def get_init_state(self, graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, randomized, batch_replication):
        self._handle = None
        self._handle_problem = None
        return self._scorer.get_init_state(graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, randomized, batch_replication)


class ReinforceDecimator(nn.Module):
    "Distributed decimation through an external force (reference: pdp_decimate.py:189-250)."

    def __init__(self, device, scorer, decimation_probability=0.5):
        super(ReinforceDecimator, self).__init__()
        self._device = device
        self._scorer = scorer
        self._decimation_probability = decimation_probability
        self._function_message_dim = 3
        self._variable_message_dim = 2
        self._handle = None
        self._handle_problem = None

    def native_handle(self, sat_problem):
        if self._handle is None or self._handle_problem is not sat_problem._native:
            self._handle = native.Decimator(sat_problem._native)
            self._handle_problem = sat_problem._native
        return self._handle

    def forward(self, init_state, message_state, sat_problem, is_training, active_mask=None):
        self.native_handle(sat_problem)
        variable_state, function_state = message_state
        coin = float(torch.rand(1).item())          # one shared coin per batch (pdp_decimate.py:218)
        am = None if active_mask is None else active_mask.reshape(-1)
        sat_problem._native.reinforce_decimate(self._handle, function_state, am, coin, self._decimation_probability, self._scorer._pi)
        return variable_state, function_state

    def get_init_state(self, graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, randomized, batch_replication):
        self._handle = None
        self._handle_problem = None
        return pdp_predict._init_sp_state(self._device, graph_map.size(1) * batch_replication, randomized)

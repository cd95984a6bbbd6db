"""Propagators of the PDP framework (reference: src/pdp/nn/pdp_propagate.py).

``SurveyPropagator`` keeps the reference's constructor and ``forward(init_state, decimator_state, sat_problem,
is_training, active_mask)`` signature (pdp_propagate.py:114-237); the sweep itself is the HIP kernel behind
``pdp_sp_propagate`` (csrc/pdp_ops.hip) instead of ten sparse products.
"""

import torch
import torch.nn as nn

from pdp import native
from pdp.nn import util


class NeuralMessagePasser(nn.Module):
    """The neural propagator: two deep-set aggregators, variables->functions and functions->variables
    (reference: pdp_propagate.py:21-108).  Each direction is one call into the matrix-core kernels."""

    def __init__(self, device, edge_dimension, decimator_dimension, meta_data_dimension, hidden_dimension, mem_hidden_dimension,
                 mem_agg_hidden_dimension, agg_hidden_dimension, dropout):
        super(NeuralMessagePasser, self).__init__()
        if edge_dimension != 1:
            raise native.NativeError("NeuralMessagePasser: edge_feature_dim = 1 only (the loader's edge feature is the literal's sign)")
        self._device = device
        self._module_list = nn.ModuleList()
        self._drop_out = dropout
        self._meta_dim = meta_data_dimension       # > 0: graph features are appended to every edge's input (pdp_propagate.py:59-61, 74-75, 85-86) -> generic operators
        self._variable_aggregator = util.MessageAggregator(device, decimator_dimension + edge_dimension + meta_data_dimension,
                                                           hidden_dimension, mem_hidden_dimension, mem_agg_hidden_dimension,
                                                           agg_hidden_dimension, edge_dimension, include_self_message=False)
        self._function_aggregator = util.MessageAggregator(device, decimator_dimension + edge_dimension + meta_data_dimension,
                                                           hidden_dimension, mem_hidden_dimension, mem_agg_hidden_dimension,
                                                           agg_hidden_dimension, edge_dimension, include_self_message=False)
        self._module_list.append(self._variable_aggregator)
        self._module_list.append(self._function_aggregator)
        self._hidden_dimension = hidden_dimension
        self._mem_hidden_dimension = mem_hidden_dimension
        self._agg_hidden_dimension = agg_hidden_dimension
        self._mem_agg_hidden_dimension = mem_agg_hidden_dimension

    def forward(self, init_state, decimator_state, sat_problem, is_training, active_mask=None):
        if util.on_train_path(self, is_training):
            return self._forward_train(init_state, decimator_state, sat_problem, active_mask)
        if self._meta_dim > 0 or sat_problem._meta_data is not None:
            # graph features: the fused kernels take [state | sign] rows only; the layers and row sums run as generic native operators
            return self._forward_train(init_state, decimator_state, sat_problem, active_mask, dropout=bool(is_training))
        if is_training and self._drop_out > 0:
            raise native.NativeError("is_training with dropout needs gradients enabled (the differentiable training path)")
        if len(decimator_state) == 3:
            decimator_variable_state, decimator_function_state, edge_mask = decimator_state
            edge_mask = edge_mask.reshape(-1).contiguous()
        else:
            decimator_variable_state, decimator_function_state = decimator_state
            edge_mask = None
        variable_state, function_state = init_state
        am = None if active_mask is None else active_mask.reshape(-1).contiguous()
        nat = sat_problem._native
        out_v, out_f = getattr(self, '_out', None) or (None, None)        # the solver's graph loop names the buffers a sweep writes
        # variables --> functions (pdp_propagate.py:72-78)
        function_state = nat.neural_aggregate_edges(self._variable_aggregator.native_weights(), True,
                                                    decimator_variable_state.contiguous(), edge_mask, am, function_state.contiguous(), out=out_f)
        # functions --> variables (pdp_propagate.py:83-89)
        variable_state = nat.neural_aggregate_edges(self._function_aggregator.native_weights(), False,
                                                    decimator_function_state.contiguous(), edge_mask, am, variable_state.contiguous(), out=out_v)
        return variable_state, function_state

    def _forward_train(self, init_state, decimator_state, sat_problem, active_mask, dropout=True):
        "the differentiable sweep of the training path (pdp_propagate.py:47-95 with is_training=True): aggregators + dropout"
        from pdp.nn import train_ops as T
        if len(decimator_state) == 3:
            dec_v, dec_f, edge_mask = decimator_state
        else:
            dec_v, dec_f = decimator_state
            edge_mask = None
        variable_state, function_state = init_state
        sign = sat_problem._edge_feature
        if active_mask is not None:
            mask = active_mask.reshape(-1).float()[sat_problem._batch_variable_map.long()][sat_problem._graph_map[0].long()].unsqueeze(1)
        gf = sat_problem.edge_meta()
        extra = (sign,) if gf is None else (sign, gf)
        # [state | sign] in front of the aggregators: the sign column travels apart (no [E, H + 1] copy) unless graph features widen the input
        if gf is None:
            fs = self._variable_aggregator.forward_train(dec_v, sign, sat_problem, True, edge_mask, state_feature=sign)
        else:
            fs = self._variable_aggregator.forward_train(torch.cat((dec_v,) + extra, 1), sign, sat_problem, True, edge_mask)
        if active_mask is not None:
            fs = mask * fs + (1 - mask) * function_state
        if dropout:
            fs = T.dropout(fs, self._drop_out, getattr(self, '_rng', 'device'))
        if gf is None:
            vs = self._function_aggregator.forward_train(dec_f, sign, sat_problem, False, edge_mask, state_feature=sign)
        else:
            vs = self._function_aggregator.forward_train(torch.cat((dec_f,) + extra, 1), sign, sat_problem, False, edge_mask)
        if active_mask is not None:
            vs = mask * vs + (1 - mask) * variable_state
        if dropout:
            vs = T.dropout(vs, self._drop_out, getattr(self, '_rng', 'device'))
        return vs, fs

    def get_init_state(self, graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, randomized, batch_replication):
        "reference: pdp_propagate.py:97-108"
        edge_num = graph_map.size(1) * batch_replication
        if randomized:
            # drawn from the global CPU generator like the reference's --cpu_mode run (default: seeded runs reproduce its numbers), or on the
            # device (config key init_rng: 'device' -- what a training run wants: 4 x [E, H] host-generated numbers per batch cost seconds)
            where = self._device if getattr(self, '_init_rng', 'torch') == 'device' else None
            variable_state = 2.0 * torch.rand(edge_num, self._hidden_dimension, dtype=torch.float32, device=where) - 1.0
            function_state = 2.0 * torch.rand(edge_num, self._hidden_dimension, dtype=torch.float32, device=where) - 1.0
            return (variable_state.to(self._device), function_state.to(self._device))
        return (torch.zeros(edge_num, self._hidden_dimension, dtype=torch.float32, device=self._device),
                torch.zeros(edge_num, self._hidden_dimension, dtype=torch.float32, device=self._device))


class SurveyPropagator(nn.Module):
    """Survey Propagation in the log domain (reference: pdp_propagate.py:114-221).  With ``include_adaptors`` (model type
    p-nd-np) two bias-free linear maps turn the neural decimator's [E, H] states into the propagator's inputs
    (pdp_propagate.py:128-131, 166-167, 179-182); the parameter names are the reference's, so its checkpoints load."""

    def __init__(self, device, decimator_dimension, include_adaptors=False, pi=0.0):
        super(SurveyPropagator, self).__init__()
        self._device = device
        self._function_message_dim = 3
        self._variable_message_dim = 2
        self._include_adaptors = include_adaptors
        self._pi = float(pi)
        if include_adaptors:
            self._variable_input_projector = nn.Linear(decimator_dimension, self._variable_message_dim, bias=False)
            self._function_input_projector = nn.Linear(decimator_dimension, 1, bias=False)
            self._module_list = nn.ModuleList([self._variable_input_projector, self._function_input_projector])

    def forward(self, init_state, decimator_state, sat_problem, is_training, active_mask=None):
        if len(decimator_state) == 3:
            dec_q, dec_fs, edge_mask = decimator_state
        else:
            dec_q, dec_fs = decimator_state
            edge_mask = None
        init_q, init_fs = init_state
        if self._include_adaptors and util.on_train_path(self, is_training):
            return self._forward_train(init_state, dec_q, dec_fs, edge_mask, sat_problem, active_mask)
        am = None if active_mask is None else active_mask.reshape(-1).contiguous()
        em = None if edge_mask is None else edge_mask.reshape(-1).contiguous()
        nat = sat_problem._native
        if self._include_adaptors:
            xlog, fs2 = nat.sp_adaptors(dec_q.contiguous(), dec_fs.contiguous(),
                                        self._function_input_projector.weight.data.reshape(-1).contiguous(),
                                        self._variable_input_projector.weight.data.contiguous())
            return nat.sp_propagate_adapted(xlog, fs2, em, am, init_q.contiguous(), init_fs.contiguous(), self._pi, out=getattr(self, '_out', None))
        return nat.sp_propagate(dec_q.contiguous(), dec_fs.contiguous(), em, am, init_q.contiguous(), init_fs.contiguous(), self._pi)

    def _forward_train(self, init_state, dec_v, dec_f, edge_mask, sat_problem, active_mask):
        """The differentiable sweep of the training path (pdp_propagate.py:139-221 with include_adaptors, under autograd): the two bias-free
        projections run on the matrix cores with their adjoints (train_ops.LinearAct), the [E, 1] / [E, 2] activations are torch element-wise
        operators, and the sweep itself is one native forward / adjoint pair (train_ops.SpAdaptedSweep).  torch.sign has no gradient, so
        only the first row of the variable projector learns from the sweep (the reference's in-place column writes, pdp_propagate.py:180-181)."""
        from pdp.nn import train_ops as T
        import torch.nn.functional as F
        xlog = F.logsigmoid(T.LinearAct.apply(dec_v, self._function_input_projector.weight, None, 'none'))
        u = T.LinearAct.apply(dec_f, self._variable_input_projector.weight, None, 'none')
        fs2 = torch.stack((torch.sigmoid(u[:, 0]), torch.sign(u[:, 1]).detach()), dim=1)
        q, fs = T.SpAdaptedSweep.apply(xlog, fs2, sat_problem._native, edge_mask, self._pi)
        if active_mask is not None:
            mask = active_mask.reshape(-1).float()[sat_problem._batch_variable_map.long()][sat_problem._graph_map[0].long()].unsqueeze(1)
            q = mask * q + (1 - mask) * init_state[0]
            fs = torch.cat((mask * fs[:, :1] + (1 - mask) * init_state[1][:, :1], fs[:, 1:]), 1)
        return q, fs

    def get_init_state(self, graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, randomized, batch_replication):
        "reference: pdp_propagate.py:223-237 (random draws come from the torch CPU generator, like the reference on CPU)"
        edge_num = graph_map.size(1) * batch_replication
        dev = self._device
        if randomized:
            where = dev if getattr(self, '_init_rng', 'torch') == 'device' else None
            variable_state = torch.rand(edge_num, self._function_message_dim, dtype=torch.float32, device=where)
            variable_state = variable_state / torch.sum(variable_state, 1).unsqueeze(1)
            function_state = torch.rand(edge_num, self._variable_message_dim, dtype=torch.float32, device=where)
            function_state[:, 1] = 0
            return (variable_state.to(dev), function_state.to(dev))
        variable_state = torch.ones(edge_num, self._function_message_dim, dtype=torch.float32, device=dev) / self._function_message_dim
        function_state = 0.5 * torch.ones(edge_num, self._variable_message_dim, dtype=torch.float32, device=dev)
        function_state[:, 1] = 0
        return (variable_state, function_state)

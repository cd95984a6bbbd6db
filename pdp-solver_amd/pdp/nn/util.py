"""Primitives of the PDP framework on the native library.

Mirrors the public names of the reference's util module for the inference path (reference:
src/pdp/nn/util.py): ``SatCNFEvaluator`` (:203-236), ``sparse_smooth_max`` / ``sparse_max`` /
``sparse_argmax`` (:257-286), ``MessageAggregator`` (:11-77), ``PerceptronTanh`` (:242-251).  The
reference evaluates these with torch sparse COO products and a dense [V x B] matrix; here each is one
kernel launch over the batch's instance-local CSR layout (see csrc/pdp_ops.hip).
``SatLossEvaluator.forward`` is differentiable with respect to the prediction (pdp/nn/train_ops.py::SatLoss); ``MultiLayerPerceptron``
is not used by any solver of the reference and is not built.
"""

import torch
import torch.nn as nn

from pdp import native



def on_train_path(module, is_training):
    """Does this call run the differentiable (autograd) path?  The solver decides ONCE per forward (PropagatorDecimatorSolverBase.forward:
    is_training, gradients enabled, a parameter that requires them) and pins the answer on its three plug-ins, so propagator, decimator and
    predictor never mix the bit-exact inference kernels with the tolerance-level training operators; a plug-in called on its own falls
    back to the same rule without the parameter test.  Inference callers pass is_training=False or run under torch.no_grad()."""
    pinned = getattr(module, '_train_path', None)
    if pinned is not None:
        return bool(pinned)
    return bool(is_training) and torch.is_grad_enabled()

class SatCNFEvaluator(nn.Module):
    """Clause-satisfaction check: returns (solved [B,1], unsat_clauses [B,1])  (reference: util.py:203-236).

    The reference rebuilds four sparse masks from ``graph_map`` on every call; the native check runs on the
    problem's resident layout, so callers that own a ``SATProblem`` should pass it as ``sat_problem``."""

    def __init__(self, device):
        super(SatCNFEvaluator, self).__init__()
        self._device = device

    def forward(self, variable_prediction, graph_map, batch_variable_map, batch_function_map, edge_feature, meta_data,
                sat_problem=None):
        if sat_problem is None:
            from pdp.nn.solver import SATProblem
            sat_problem = SATProblem((graph_map, batch_variable_map, batch_function_map, edge_feature, meta_data, None),
                                     self._device, 1)
        handle = sat_problem._native if sat_problem._batch_replication == 1 else sat_problem._native_unreplicated()
        return handle.cnf_eval(variable_prediction.reshape(-1).contiguous())


class SatLossEvaluator(nn.Module):
    """Energy of a prediction = the training loss, reported by the test mode (reference: util.py:110-197).  ``forward`` keeps the
    reference's argument list; the computation is one native call on the problem's resident layout (pass ``sat_problem``), where the
    reference rebuilds two sparse masks and runs three sparse products per call."""

    def __init__(self, alpha, device):
        super(SatLossEvaluator, self).__init__()
        self._alpha = alpha
        self._device = device

    def forward(self, variable_prediction, label, graph_map, batch_variable_map, batch_function_map, edge_feature, meta_data,
                global_step, eps, max_coeff, loss_sharpness, sat_problem=None):
        if float(loss_sharpness) != int(loss_sharpness) or int(loss_sharpness) < 1:
            raise native.NativeError("SatLossEvaluator: the native kernel takes a positive integer loss_sharpness")
        if sat_problem is None:
            from pdp.nn.solver import SATProblem
            sat_problem = SATProblem((graph_map, batch_variable_map, batch_function_map, edge_feature, meta_data, None),
                                     self._device, 1)
        handle = sat_problem._native if sat_problem._batch_replication == 1 else sat_problem._native_unreplicated()
        gs = global_step.detach().to(torch.float32).reshape(-1)[:1].cpu()
        coeff = float(torch.min(gs.pow(self._alpha), torch.tensor([float(max_coeff)])).item())     # util.py:181
        e = float(eps.reshape(-1)[0].item()) if torch.is_tensor(eps) else float(eps)
        if torch.is_grad_enabled() and variable_prediction.requires_grad:
            from pdp.nn import train_ops as T
            return T.SatLoss.apply(variable_prediction, handle, coeff, e, int(loss_sharpness))
        return handle.sat_loss(variable_prediction.reshape(-1).contiguous(), coeff, e, int(loss_sharpness)).reshape(())


class MessageAggregator(nn.Module):
    """Deep-set message aggregation at variable / function nodes (reference: util.py:11-77).

    Same parameters and state-dict keys as the reference (``_W1_m``, ``_W2_m``, ``_W1_a``, ``_W2_a`` plus their aliases in
    ``_module_list``), so reference checkpoints load unchanged.  The computation runs on the fp32 matrix cores
    (csrc/pdp_neural.hip); the module only keeps padded, transposed copies of its weights for the kernels."""

    def __init__(self, device, input_dimension, output_dimension, mem_hidden_dimension,
                 mem_agg_hidden_dimension, agg_hidden_dimension, feature_dimension, include_self_message):
        super(MessageAggregator, self).__init__()
        if not (mem_hidden_dimension > 0 and mem_agg_hidden_dimension > 0 and agg_hidden_dimension > 0):
            raise native.NativeError("MessageAggregator: the native kernels implement the full 4-layer form "
                                     "(mem_hidden_dim, mem_agg_hidden_dim, agg_hidden_dim > 0) used by every reference config")
        self._device = device
        self._include_self_message = include_self_message
        self._module_list = nn.ModuleList()
        self._W1_m = nn.Linear(input_dimension, mem_hidden_dimension, bias=True)
        self._W2_m = nn.Linear(mem_hidden_dimension, mem_agg_hidden_dimension, bias=False)
        self._module_list.append(self._W1_m)
        self._module_list.append(self._W2_m)
        self._W1_a = nn.Linear(mem_agg_hidden_dimension + feature_dimension, agg_hidden_dimension, bias=True)
        self._W2_a = nn.Linear(agg_hidden_dimension, output_dimension, bias=False)
        self._module_list.append(self._W1_a)
        self._module_list.append(self._W2_a)
        self._agg_hidden_dimension = agg_hidden_dimension
        self._mem_hidden_dimension = mem_hidden_dimension
        self._mem_agg_hidden_dimension = mem_agg_hidden_dimension
        self._feature_dimension = feature_dimension
        self._native = None
        self._native_key = None

    def native_weights(self):
        "padded / transposed device copies, rebuilt whenever a parameter was modified in place or replaced"
        params = (self._W1_m.weight, self._W1_m.bias, self._W2_m.weight, self._W1_a.weight, self._W1_a.bias, self._W2_a.weight)
        key = tuple((p.data_ptr(), p._version) for p in params)
        if self._native is None or key != self._native_key:
            self._native = native.AggregatorWeights(*[p.data for p in params], feature_dim=self._feature_dimension)
            self._native_key = key
        return self._native


    def forward_train(self, state, feature, sat_problem, by_variable, edge_mask=None):
        """The differentiable form (training; reference: util.py:51-77): ``state`` [E, input_dimension] already carries the appended edge
        feature, ``feature`` [E, feature_dimension] (or None) is appended after the aggregation.  Every layer and the row aggregation is a
        native forward / adjoint pair (pdp/nn/train_ops.py)."""
        from pdp.nn import train_ops as T
        s = T.LinearAct.apply(state, self._W1_m.weight, self._W1_m.bias, 'logsigmoid')
        s = T.LinearAct.apply(s, self._W2_m.weight, None, 'logsigmoid')
        if edge_mask is not None:
            s = s * edge_mask
        agg = T.RowAggregate.apply(s, sat_problem._native, by_variable, self._include_self_message)
        if feature is not None:
            agg = torch.cat((agg, feature), 1)
        g = T.LinearAct.apply(agg, self._W1_a.weight, self._W1_a.bias, 'logsigmoid')
        return T.LinearAct.apply(g, self._W2_a.weight, None, 'logsigmoid')


class PerceptronTanh(nn.Module):
    "1-hidden-layer perceptron with tanh output (reference: util.py:242-251)."

    def __init__(self, input_dimension, hidden_dimension, output_dimension):
        super(PerceptronTanh, self).__init__()
        self._layer1 = nn.Linear(input_dimension, hidden_dimension)
        self._layer2 = nn.Linear(hidden_dimension, output_dimension, bias=False)

    def forward(self, inp):
        return torch.tanh(self._layer2(torch.relu(self._layer1(inp))))


def sparse_smooth_max(x, sat_problem, device=None, alpha=30):
    """Per-variable smooth max of an edge vector ``x [E,1]`` (reference: util.py:282-286).  The reference takes
    the sparse variable mask; the native form takes the problem that owns it."""
    if alpha != 30:
        raise native.NativeError("sparse_smooth_max: the native kernel implements alpha = 30 (the only value the reference uses)")
    return sat_problem._native.smooth_max(x.reshape(-1).contiguous())


def sparse_max(x, sat_problem, device=None):
    "Exact per-instance max of a variable vector ``x [V]`` incl. the reference's x - min + 1 rounding (util.py:267-275)."
    return sat_problem._native.instance_max(x.reshape(-1).contiguous())


def sparse_argmax(x, sat_problem, device=None):
    "Per-instance arg-max (global variable index, first index wins ties) of ``x [V]`` (util.py:257-265)."
    return sat_problem._native.instance_argmax(x.reshape(-1).contiguous())

"""Primitives of the PDP framework on the native library.

Mirrors the public names of the reference's util module for the inference path (reference:
src/pdp/nn/util.py): ``SatCNFEvaluator`` (:203-236), ``sparse_smooth_max`` / ``sparse_max`` /
``sparse_argmax`` (:257-286), ``MessageAggregator`` (:11-77), ``PerceptronTanh`` (:242-251).  The
reference evaluates these with torch sparse COO products and a dense [V x B] matrix; here each is one
kernel launch over the batch's instance-local CSR layout (see csrc/pdp_ops.hip).
``SatLossEvaluator.forward`` is differentiable with respect to the prediction (pdp/nn/train_ops.py::SatLoss); ``MultiLayerPerceptron``
is not used by any solver of the reference and is not built.

Call shapes are the reference's: ``sparse_max(x, mask, device)``, ``sparse_argmax(x, mask, device)``, ``sparse_smooth_max(x, mask, device,
alpha)``, ``safe_exp(x, device)``, ``MessageAggregator.forward(state, feature, mask, mask_transpose, edge_mask)`` take torch sparse masks.
A mask that ``SATProblem`` built (``sat_problem._graph_mask_tuple`` ...) carries a tag naming its problem and is mapped back to the
resident layout, i.e. to the same kernels the native plug-ins use; any other sparse mask runs on the generic index-list kernels of
csrc/pdp_coo.hip.  Either way the arithmetic is native -- a plug-in written against the reference keeps working unchanged
(tests/golden/foreign_plugin.py, tests/test_foreign_plugin.py).
"""

import weakref

import torch
import torch.nn as nn

from pdp import native


# ---- sparse masks: the ones SATProblem built are tagged, the rest go through their index lists ----------------------------------------
def tag_mask(mask, sat_problem, kind, slot):
    "called by SATProblem for every mask it materialises: remembers which problem / tuple / position the tensor is"
    mask._pdp_tag = (weakref.ref(sat_problem), kind, slot)
    return mask


def mask_owner(mask):
    "(sat_problem, kind, slot) of a mask built by a live SATProblem, else None"
    tag = getattr(mask, '_pdp_tag', None)
    if tag is None:
        return None
    problem = tag[0]()
    return None if problem is None else (problem, tag[1], tag[2])


def _csr(mask):
    "row-sorted form of an arbitrary sparse mask, built once and kept on the tensor"
    c = getattr(mask, '_pdp_csr', None)
    if c is None:
        if not mask.is_sparse:
            raise native.NativeError("a sparse (COO) mask is expected, got a dense tensor")
        c = native.CsrMask(mask)
        mask._pdp_csr = c
    return c


def _is_problem(obj):
    from pdp.nn.solver import SATProblem
    return isinstance(obj, SATProblem)



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
        from pdp.nn.solver import SATProblem
        if sat_problem is None:
            # the reference's call shape (trainer.py:113-115, :134-136, :153-155): no problem object, only its tensors.  A problem that is
            # still alive and owns exactly these tensors is found again (no rebuild per sweep in a reference-shaped termination callback)
            handle = SATProblem.handle_of(graph_map, variable_prediction.numel())
            if handle is None:
                handle = SATProblem((graph_map, batch_variable_map, batch_function_map, edge_feature, meta_data, None), self._device, 1)._native
        elif variable_prediction.numel() == sat_problem._variable_num:
            handle = sat_problem._native
        else:
            handle = sat_problem._native_unreplicated()            # a de-duplicated prediction of a replicated batch
        return handle.cnf_eval(variable_prediction.reshape(-1).contiguous())


class SatLossEvaluator(nn.Module):
    """Energy of a prediction = the training loss, reported by the test mode (reference: util.py:110-197).  ``forward`` keeps the
    reference's argument list; the computation is one native call on the problem's resident layout (pass ``sat_problem``), where the
    reference rebuilds two sparse masks and runs three sparse products per call."""

    def __init__(self, alpha, device):
        super(SatLossEvaluator, self).__init__()
        self._alpha = alpha
        self._device = device

    @staticmethod
    def safe_log(x, eps):
        "log(max(x, eps)) (util.py:121-123)"
        return torch.max(x, eps).log()

    @staticmethod
    def _coo(row_index, col_index, values, shape, device):
        return torch.sparse_coo_tensor(torch.stack([row_index.long(), col_index.long()]), values, torch.Size(shape), device=device)

    @staticmethod
    def compute_masks(graph_map, batch_variable_map, batch_function_map, edge_feature, device):
        """(signed edge-by-variable mask [E, V], clause-by-edge mask [F, E]) as torch sparse tensors (util.py:125-148).  The native
        evaluators do not need them; they are built for callers written against the reference."""
        E, V, F = graph_map.size(1), batch_variable_map.size(0), batch_function_map.size(0)
        edge_ids = torch.arange(E, dtype=torch.int64, device=device)
        variable_mask = SatLossEvaluator._coo(edge_ids, graph_map[0, :], edge_feature.squeeze(1), (E, V), device)
        function_mask = SatLossEvaluator._coo(graph_map[1, :], edge_ids, torch.ones(E, device=device), (F, E), device)
        return variable_mask, function_mask

    @staticmethod
    def compute_batch_mask(batch_variable_map, batch_function_map, device):
        "(variable-by-instance mask [V, B], its transpose, clause-by-instance mask [F, B], its transpose) (util.py:150-176)"
        V, F = batch_variable_map.size(0), batch_function_map.size(0)
        B = int((batch_variable_map.max() + 1).long().item())
        variable_mask = SatLossEvaluator._coo(torch.arange(V, dtype=torch.int64, device=device), batch_variable_map,
                                              torch.ones(V, device=device), (V, B), device)
        function_mask = SatLossEvaluator._coo(torch.arange(F, dtype=torch.int64, device=device), batch_function_map,
                                              torch.ones(F, device=device), (F, B), device)
        return (variable_mask, variable_mask.transpose(0, 1), function_mask, function_mask.transpose(0, 1))

    def forward(self, variable_prediction, label, graph_map, batch_variable_map, batch_function_map, edge_feature, meta_data,
                global_step, eps, max_coeff, loss_sharpness, sat_problem=None):
        if float(loss_sharpness) != int(loss_sharpness) or int(loss_sharpness) < 1:
            raise native.NativeError("SatLossEvaluator: the native kernel takes a positive integer loss_sharpness")
        from pdp.nn.solver import SATProblem
        if sat_problem is None:
            handle = SATProblem.handle_of(graph_map, variable_prediction.numel())
            if handle is None:
                handle = SATProblem((graph_map, batch_variable_map, batch_function_map, edge_feature, meta_data, None), self._device, 1)._native
        elif variable_prediction.numel() == sat_problem._variable_num:
            handle = sat_problem._native
        else:
            handle = sat_problem._native_unreplicated()
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
        self._device = device
        self._include_self_message = include_self_message
        self._module_list = nn.ModuleList()
        # either pair of layers is optional (util.py:24-42): without the first pair the raw state is aggregated
        self._has_pre = mem_hidden_dimension > 0 and mem_agg_hidden_dimension > 0
        self._has_post = agg_hidden_dimension > 0 and mem_agg_hidden_dimension > 0
        if self._has_pre:
            self._W1_m = nn.Linear(input_dimension, mem_hidden_dimension, bias=True)
            self._W2_m = nn.Linear(mem_hidden_dimension, mem_agg_hidden_dimension, bias=False)
            self._module_list.append(self._W1_m)
            self._module_list.append(self._W2_m)
        if self._has_post:
            if mem_hidden_dimension <= 0:
                mem_agg_hidden_dimension = input_dimension
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
        if not (self._has_pre and self._has_post):
            raise native.NativeError("MessageAggregator: the fused kernels implement the full 4-layer form (mem_hidden_dim, mem_agg_hidden_dim, "
                                     "agg_hidden_dim > 0) of every reference config; call the module itself for the partial forms")
        params = (self._W1_m.weight, self._W1_m.bias, self._W2_m.weight, self._W1_a.weight, self._W1_a.bias, self._W2_a.weight)
        key = tuple((p.data_ptr(), p._version) for p in params)
        if self._native is None or key != self._native_key:
            self._native = native.AggregatorWeights(*[p.data for p in params], feature_dim=self._feature_dimension)
            self._native_key = key
        return self._native


    def forward(self, state, feature, mask, mask_transpose, edge_mask=None):
        """The reference's call (util.py:51-77): ``state`` [E, input_dimension], ``feature`` [rows, feature_dimension] or None, ``mask`` [rows, E]
        and ``mask_transpose`` [E, rows] sparse, ``edge_mask`` [E, 1] or None.  Returns [rows, out] with include_self_message, else [E, out].
        Layers run on the matrix cores (pdp_train_linear), the aggregation on the resident CSR layout when ``mask`` is a variable / clause
        mask of a SATProblem (pdp_train_row_sum / _spread) and on the mask's own index lists otherwise (pdp_csr_matmul).  Differentiable."""
        native.require_gpu()
        owner = mask_owner(mask)
        if owner is not None and owner[1] == 'graph' and owner[2] in (0, 2) and state.size(0) == owner[0]._edge_num:
            return self.forward_train(state, feature, owner[0], owner[2] == 0, edge_mask)
        return self.forward_train(state, feature, None, None, edge_mask, masks=(mask, mask_transpose))

    def forward_train(self, state, feature, sat_problem, by_variable, edge_mask=None, masks=None, state_feature=None):
        """The differentiable form (training; reference: util.py:51-77): ``state`` [E, input_dimension] already carries the appended edge
        feature -- or ``state`` is [E, input_dimension - 1] and ``state_feature`` [E, 1] is that last column held apart (no concatenation is
        made where the row-stripe GEMM takes the layer, train_ops.linear_sign); ``feature`` [E, feature_dimension] (or None) is appended
        after the aggregation, the same way.  Every layer and the row aggregation is a native forward / adjoint pair (pdp/nn/train_ops.py).
        ``masks`` = (mask, mask_transpose) replaces the problem's own rows."""
        from pdp.nn import train_ops as T
        s = state
        if self._has_pre:
            if state_feature is not None:
                s = T.linear_sign(s, state_feature, self._W1_m.weight, self._W1_m.bias, 'logsigmoid')
            else:
                s = T.LinearAct.apply(s, self._W1_m.weight, self._W1_m.bias, 'logsigmoid')
            s = T.LinearAct.apply(s, self._W2_m.weight, None, 'logsigmoid')
        elif state_feature is not None:
            s = torch.cat((s, state_feature), 1)
        if edge_mask is not None:
            s = s * edge_mask
        if masks is None:
            agg = T.RowAggregate.apply(s, sat_problem._native, by_variable, self._include_self_message)
        else:
            agg = T.MaskMatmul.apply(s, masks[0])
            if not self._include_self_message:
                agg = T.MaskMatmul.apply(agg, masks[1]) - (s * edge_mask if edge_mask is not None else s)
        if self._has_post:
            if feature is not None and feature.dim() == 2 and feature.size(1) == 1:
                agg = T.linear_sign(agg, feature, self._W1_a.weight, self._W1_a.bias, 'logsigmoid')
            else:
                if feature is not None:
                    agg = torch.cat((agg, feature), 1)
                agg = T.LinearAct.apply(agg, self._W1_a.weight, self._W1_a.bias, 'logsigmoid')
            agg = T.LinearAct.apply(agg, self._W2_a.weight, None, 'logsigmoid')
        elif feature is not None:
            agg = torch.cat((agg, feature), 1)
        return agg


class PerceptronTanh(nn.Module):
    "1-hidden-layer perceptron with tanh output (reference: util.py:242-251)."

    def __init__(self, input_dimension, hidden_dimension, output_dimension):
        super(PerceptronTanh, self).__init__()
        self._layer1 = nn.Linear(input_dimension, hidden_dimension)
        self._layer2 = nn.Linear(hidden_dimension, output_dimension, bias=False)

    def forward(self, inp):
        return torch.tanh(self._layer2(torch.relu(self._layer1(inp))))


def sparse_argmax(x, mask, device=None):
    """For every column of ``mask`` [rows, groups] the row index of its largest x (first index on ties, 0 for an empty column), computed like
    the reference's dense arg-max over ``x - x.min() + 1`` (util.py:257-265).  ``x`` [nnz] is paired with the mask's entries in index order.
    The variable-by-instance mask of a SATProblem (``_batch_mask_tuple[0]``) maps to pdp_instance_argmax on the resident layout."""
    x = x.reshape(-1).contiguous()
    if _is_problem(mask):
        return mask._native.instance_argmax(x)
    owner = mask_owner(mask)
    if owner is not None and owner[1] == 'batch' and owner[2] == 0:
        return owner[0]._native.instance_argmax(x)
    idx = mask._indices()
    return native.coo_reduce('argmax', idx[0].contiguous(), idx[1].contiguous(), x, int(mask.size(0)), int(mask.size(1)))


def sparse_max(x, mask, device=None):
    "Exact per-column max of x over the entries of ``mask``, incl. the reference's ``x - min + 1`` rounding (util.py:267-275)."
    x = x.reshape(-1).contiguous()
    if _is_problem(mask):
        return mask._native.instance_max(x)
    owner = mask_owner(mask)
    if owner is not None and owner[1] == 'batch' and owner[2] == 0:
        return owner[0]._native.instance_max(x)
    idx = mask._indices()
    return native.coo_reduce('max', idx[0].contiguous(), idx[1].contiguous(), x, int(mask.size(0)), int(mask.size(1)))


def safe_exp(x, device=None):
    "exp(min(x, 30)) (util.py:277-280)"
    return native.math_apply('safe_exp', x.contiguous())


def sparse_smooth_max(x, mask, device=None, alpha=30):
    """Soft-max-weighted mean of ``x`` [E, 1] over the entries of every row of ``mask`` [rows, E]:
    mm(mask, x * c) / max(mm(mask, c), 1) with c = exp(min(alpha * x, 30)) (util.py:282-286).  Returns [rows, 1].  The edge-by-variable mask
    of a SATProblem (``_graph_mask_tuple[0]``) with the reference's alpha = 30 maps to pdp_smooth_max on the resident layout."""
    x = x.reshape(-1).contiguous()
    if _is_problem(mask):
        owner = (mask, 'graph', 0)
    else:
        owner = mask_owner(mask)
    if owner is not None and owner[1] == 'graph' and owner[2] == 0 and float(alpha) == 30.0:
        return owner[0]._native.smooth_max(x)
    return _csr(mask).smooth_max(x, float(alpha))

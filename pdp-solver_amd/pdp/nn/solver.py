"""The PDP solver framework on the MI355X-native library.

Keeps the reference's class names, constructor arguments and ``forward`` / ``get_init_state`` signatures
(reference: src/pdp/nn/solver.py) so that ``trainer`` / ``satyr.py`` and third-party plug-ins keep working:

* ``SATProblem`` (solver.py:19-285) -- batch container.  The reference materialises 24 sparse COO masks; here the
  batch lives in an instance-local CSR layout in HBM (csrc/pdp_problem.hip) and the reference's mask tuples are
  only built lazily for plug-ins that still want them.
* ``PropagatorDecimatorSolverBase`` (solver.py:293-511) -- the PDP loop.  When the (propagator, decimator,
  predictor, termination check) quadruple is the native classical one, the whole ``_forward_core`` loop runs as ONE
  persistent kernel (csrc/pdp_solve.hip); otherwise the generic step-wise loop below calls the plug-ins exactly like
  the reference does.  Results are identical either way (tests/test_api_forward.py).
"""

import os
import weakref

import numpy as np
import torch
import torch.nn as nn

from pdp import native
from pdp.nn import pdp_propagate, pdp_decimate, pdp_predict, util


###############################################################
### The Problem Class
###############################################################

class SATProblem(object):
    "A batch of CNF instances resident in HBM (reference: solver.py:19-285)."

    _live = weakref.WeakValueDictionary()         # id(graph tensor) -> problem that owns it (see handle_of)

    @classmethod
    def handle_of(cls, graph_map, variable_count):
        """The native handle of the live problem that owns ``graph_map`` (its ``_graph_map`` attribute, or the tensor it was built from) and
        has ``variable_count`` variables: lets the evaluators, which the reference calls with bare tensors (trainer.py:153-155), reuse the
        resident batch instead of rebuilding it.  None if there is no such problem."""
        p = cls._live.get(id(graph_map))
        if p is None:
            return None
        if graph_map is p._graph_map and variable_count == p._variable_num:
            return p._native
        if graph_map is p._orig[0] and variable_count * p._batch_replication == p._variable_num:
            return p._native if p._batch_replication == 1 else p._native_unreplicated()
        return None

    def __init__(self, data_batch, device, batch_replication=1):
        self._device = device
        self._batch_replication = batch_replication
        graph_map, batch_variable_map, batch_function_map, edge_feature, meta_data, _ = data_batch
        self._orig = (graph_map, batch_variable_map, batch_function_map, edge_feature)
        self._native = native.Problem(graph_map, batch_variable_map, batch_function_map, edge_feature,
                                      replication=batch_replication)
        self._native1 = None
        # per-instance graph features [B0, M] (solver.py:35; replicated like the batch, :76-77).  No shipped config has them
        # (has_meta_data is False, trainer.py:39, and the loader yields None); the neural plug-ins take them on their generic operators.
        self._meta_data = None
        if meta_data is not None:
            self._meta_data = meta_data.to(torch.float32)
            if batch_replication > 1:
                self._meta_data = self._meta_data.repeat(batch_replication, 1)
        self._edge_meta = None
        self._variable_num = self._native.V
        self._function_num = self._native.F
        self._edge_num = self._native.E
        self._batch_size = self._native.B
        if batch_replication > 1:
            self._graph_map, self._batch_variable_map, self._batch_function_map, self._edge_feature = self._native.export_graph()
        else:
            self._graph_map, self._batch_variable_map, self._batch_function_map = graph_map, batch_variable_map, batch_function_map
            self._edge_feature = edge_feature.reshape(-1, 1)
        # state tensors are owned here and mutated in place by the kernels (solver.py:49-54)
        self._active_variables = self._native.active_variables
        self._active_functions = self._native.active_functions
        self._solution = self._native.solution
        self._is_sat = self._native.is_sat
        self._edge_mask = None
        self._masks = {}
        SATProblem._live[id(self._graph_map)] = self
        SATProblem._live[id(graph_map)] = self

    def edge_meta(self):
        """[E, M]: the graph features of every edge's instance -- what the reference computes as
        mm(variable_mask_transpose, mm(b_variable_mask, meta_data)) (pdp_propagate.py:59-61), a gather here; None without meta data"""
        if self._meta_data is None:
            return None
        if self._edge_meta is None:
            inst = self._batch_variable_map.long()[self._graph_map[0].long()]
            self._edge_meta = self._meta_data[inst].contiguous()
        return self._edge_meta

    def _native_unreplicated(self):
        "handle on the original (non-replicated) batch, used by evaluators that look at de-duplicated predictions"
        if self._native1 is None:
            gm, bvm, bfm, ef = self._orig
            self._native1 = native.Problem(gm, bvm, bfm, ef, replication=1)
        return self._native1

    # ---- K7 / K8 -------------------------------------------------------------------------------------------
    def simplify(self):
        "unit propagation + pure-literal elimination to a fix-point (solver.py:281-285)"
        self._native.simplify()

    def set_variables(self, assignment):
        "fix variables ({-1,0,+1} per variable) and simplify (solver.py:275-279); ``assignment`` is masked in place"
        self._native.set_variables(assignment.reshape(-1))

    def refresh_edge_mask(self):
        "solver.py:370-371; returns True when every edge is still active"
        all_active = self._native.refresh_edge_mask(True)
        self._edge_mask = self._native.edge_mask
        return all_active

    # ---- lazily materialised reference masks (only for third-party plug-ins) --------------------------------
    def _sparse(self, rows, cols, vals, shape):
        return torch.sparse_coo_tensor(torch.stack([rows.long(), cols.long()]), vals, shape, device=self._device)

    def _mask_tuple(self, kind):
        """The reference's mask tuples (solver.py:84-178) as real torch sparse tensors, built on first use.  Each tensor is tagged with
        (problem, tuple, position) so that pdp.nn.util maps it back to the resident layout when a plug-in hands it to sparse_max /
        sparse_argmax / sparse_smooth_max / MessageAggregator; torch.mm on them works as in the reference."""
        if kind in self._masks:
            return self._masks[kind]
        if kind == 'replication':
            # solver.py:84-99: replica r of instance i is row i + r * B0; None without replication (the reference has no attribute then;
            # its _deduplicate tests `_batch_replication <= 1` first, solver.py:404)
            out = None
            if self._batch_replication > 1:
                B, R = self._batch_size, self._batch_replication
                b0 = B // R
                m = self._sparse(torch.arange(B, device=self._device), torch.arange(b0, device=self._device).repeat(R),
                                 torch.ones(B, device=self._device), (B, b0))
                out = (util.tag_mask(m, self, kind, 0), util.tag_mask(m.transpose(0, 1), self, kind, 1))
            self._masks[kind] = out
            return out
        gm, E, V, F, B = self._graph_map, self._edge_num, self._variable_num, self._function_num, self._batch_size
        er = torch.arange(E, device=self._device)
        ef = self._edge_feature.reshape(-1)
        if kind == 'graph' or kind in ('pos', 'neg', 'signed'):
            vals = {'graph': torch.ones(E, device=self._device), 'pos': (ef == 1).float(), 'neg': (ef == -1).float(),
                    'signed': ef}[kind]
            vm = self._sparse(gm[0], er, vals, (V, E)); fm = self._sparse(gm[1], er, vals, (F, E))
            out = (vm, vm.transpose(0, 1), fm, fm.transpose(0, 1))
        elif kind == 'batch':
            vm = self._sparse(torch.arange(V, device=self._device), self._batch_variable_map, torch.ones(V, device=self._device), (V, B))
            fm = self._sparse(torch.arange(F, device=self._device), self._batch_function_map, torch.ones(F, device=self._device), (F, B))
            out = (vm, vm.transpose(0, 1), fm, fm.transpose(0, 1))
        elif kind == 'vf':
            m = self._sparse(gm[0], gm[1], torch.ones(E, device=self._device), (V, F))
            sm = self._sparse(gm[0], gm[1], ef, (V, F))
            out = (m, m.transpose(0, 1), sm, sm.transpose(0, 1))
        else:
            raise KeyError(kind)
        out = tuple(util.tag_mask(t, self, kind, i) for i, t in enumerate(out))
        self._masks[kind] = out
        return out

    _graph_mask_tuple = property(lambda self: self._mask_tuple('graph'))
    _pos_mask_tuple = property(lambda self: self._mask_tuple('pos'))
    _neg_mask_tuple = property(lambda self: self._mask_tuple('neg'))
    _signed_mask_tuple = property(lambda self: self._mask_tuple('signed'))
    _batch_mask_tuple = property(lambda self: self._mask_tuple('batch'))
    _vf_mask_tuple = property(lambda self: self._mask_tuple('vf'))
    _replication_mask_tuple = property(lambda self: self._mask_tuple('replication'))


###############################################################
### The Solver Classes
###############################################################

class OwnedState(list):
    """An initial state whose ownership travels with the call: ``model.forward(init_state=OwnedState(model.get_init_state(...)), ...)``.
    A Python caller keeps its arguments alive for the whole call (they stay on its evaluation stack), so a plain tuple pins the four
    [E, H] initial tensors of a neural solver until ``forward`` returns -- 52 GB on the configs[3] shard -- although they are dead after the
    first sweep.  ``forward`` empties an OwnedState as soon as it has unpacked it; what the caller still holds is an empty list.  The
    predict / test drivers use it; a plain tuple works as in the reference (and is kept alive by its owner, as in the reference)."""

    def take(self):
        items = tuple(self)
        self.clear()
        return items


def _is_standard_termination(check_termination):
    "the trainer's CNF-check callback (trainer.py:150-162) is implemented inside the persistent kernel"
    return check_termination is not None and getattr(check_termination, '_pdp_standard_termination', False)


class PropagatorDecimatorSolverBase(nn.Module):
    "The base class for all PDP SAT solvers (reference: solver.py:293-511)."

    def __init__(self, device, name, propagator, decimator, predictor, local_search_iterations=0, epsilon=0.05,
                 rng='torch', seed=0, persistent=True):
        super(PropagatorDecimatorSolverBase, self).__init__()
        self._device = device
        self._module_list = nn.ModuleList()
        self._propagator = propagator
        self._decimator = decimator
        self._predictor = predictor
        self._module_list.append(self._propagator)
        self._module_list.append(self._decimator)
        self._module_list.append(self._predictor)
        self._global_step = nn.Parameter(torch.tensor([0], dtype=torch.float, device=self._device), requires_grad=False)
        self._name = name
        self._local_search_iterations = local_search_iterations
        self._epsilon = epsilon
        self._rng = rng                  # 'torch': reference-compatible CPU generator stream, 'philox': on device
        self._seed = seed
        self._persistent = persistent    # allow the one-launch persistent loop when the plug-ins permit it
        self._exchange = None            # callable: this process solves a PART of every forward, the batch-wide reductions are completed by it (pdp/parallel.py)
        self._isolated = False           # True: "fixed" semantics -- instances solved on their own, no batch-wide couplings (persistent loop only)
        self.last_run = {}               # diagnostics of the most recent forward (iterations, path taken ...)

    def parameter_count(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def set_random_key(self, seed, first_variable=0, first_instance=0):
        """Key of the device-side (Philox) random numbers of the next forward: the predict driver derives it from the run's seed and the
        global (loader batch, segment) index, so a batch draws the same numbers on whichever rank it is solved.  ``first_variable`` /
        ``first_instance``: the next forward solves a contiguous PART of that segment (isolated instances dealt to ranks,
        pdp/parallel.py) which starts there -- its variables and instances draw what they draw when the segment is solved whole."""
        self._seed = int(seed)
        self._rng_base = (int(first_variable), int(first_instance))
        if hasattr(self._predictor, '_seed'):
            self._predictor._seed = int(seed)

    def save(self, export_path_base):
        torch.save(self.state_dict(), os.path.join(export_path_base, self._name))

    def load(self, import_path_base):
        self.load_state_dict(torch.load(os.path.join(import_path_base, self._name), map_location=self._device))

    # -----------------------------------------------------------------------------------------------------------
    def forward(self, init_state, graph_map, batch_variable_map, batch_function_map, edge_feature,
                meta_data, is_training=True, iteration_num=1, check_termination=None, simplify=True, batch_replication=1):
        native.require_gpu()
        # The initial state is only needed until the first sweep has read it.  It travels down in an OwnedState box that the frame which
        # finally uses it empties: no frame on the way keeps a reference, and a caller that hands over an OwnedState itself -- the predict /
        # test drivers do -- lets the four [E, H] tensors go after the first sweep (52 GB on the configs[3] shard).  A caller that keeps
        # a plain tuple keeps the tensors, as with the reference.
        owned = OwnedState(init_state.take() if isinstance(init_state, OwnedState) else init_state)
        del init_state
        batch_replication = 1 if is_training else batch_replication
        sat_problem = SATProblem((graph_map, batch_variable_map, batch_function_map, edge_feature, meta_data, None),
                                 self._device, batch_replication)
        if getattr(self, '_rng_base', (0, 0)) != (0, 0):
            sat_problem._native.set_rng_base(*self._rng_base)
        if self._exchange is not None:
            if is_training or batch_replication != 1:
                raise native.NativeError("a forward spread over several processes: prediction without batch replication only")
            sat_problem._native.set_exchange(self._exchange)
        self.last_run = dict(path='none', iterations=0, walksat_steps=0)
        # one decision for the whole forward: the differentiable operators (tolerance-level, autograd graph kept) only when training was
        # asked for, gradients are enabled and some parameter wants them; otherwise every plug-in runs its fused inference kernels
        train_path = bool(is_training) and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        for plug_in in (self._propagator, self._decimator, self._predictor):
            if plug_in is not None:
                plug_in._train_path = train_path
        self.last_run['train_path'] = train_path

        try:
            if simplify and not is_training:
                sat_problem.simplify()

            if self._propagator is not None and self._decimator is not None:
                propagator_state, decimator_state = self._forward_core(owned, None, sat_problem, iteration_num, is_training, check_termination)
            else:
                decimator_state = None
                propagator_state = None

            prediction = self._predictor(decimator_state, sat_problem, True)
        finally:
            # the pin lives for this forward only: a plug-in called on its own afterwards decides by is_training / grad mode again
            for plug_in in (self._propagator, self._decimator, self._predictor):
                if plug_in is not None:
                    plug_in._train_path = None

        if not is_training:
            prediction = self._local_search(prediction, sat_problem, batch_replication)

        prediction = self._update_solution(prediction, sat_problem)

        if batch_replication > 1:
            prediction, propagator_state, decimator_state = self._deduplicate(prediction, propagator_state, decimator_state, sat_problem)

        self._last_problem = sat_problem
        return (prediction, (propagator_state, decimator_state))

    # -----------------------------------------------------------------------------------------------------------
    @staticmethod
    def _replicas_identical(sat_problem, states):
        """Batch replication with a deterministic initial state (the predict path, base.py:288): every replica of an instance starts from
        the same messages, so the R copies run through identical sweeps and the replica-aware termination rule (trainer.py:157-160: all
        copies stop when one is solved) coincides with the per-copy rule the persistent kernel implements."""
        R = sat_problem._batch_replication
        for x in states:
            v = x.reshape(R, -1)
            if not bool((v == v[0:1]).all().item()):
                return False
        return True

    def _persistent_model(self):
        "which triple of the persistent kernel this solver is: MODEL_SP, MODEL_REINFORCE or None (plug-ins of other types: generic loop)"
        if (type(self._propagator) is not pdp_propagate.SurveyPropagator or self._propagator._include_adaptors
                or type(getattr(self._decimator, '_scorer', None)) is not pdp_predict.SurveyScorer
                or self._propagator._pi != self._decimator._scorer._pi):
            return None
        if type(self._decimator) is pdp_decimate.SequentialDecimator and type(self._predictor) is pdp_predict.IdentityPredictor:
            return native.MODEL_SP
        if type(self._decimator) is pdp_decimate.ReinforceDecimator and type(self._predictor) is pdp_predict.ReinforcePredictor:
            return native.MODEL_REINFORCE
        return None

    def _can_run_persistent(self, sat_problem, is_training, check_termination, states=()):
        model = self._persistent_model()
        if model is None or not self._persistent or is_training or not _is_standard_termination(check_termination):
            return False
        if model == native.MODEL_REINFORCE:
            # the Reinforce gate exists only with a termination callback (active_mask is None otherwise, pdp_decimate.py:205) and while
            # the batch has an active variable (a batch-wide condition that the loop cannot change: nothing is decimated)
            if check_termination is None or self._isolated or not bool((sat_problem._active_variables > 0).any().item()):
                return False
        if sat_problem._batch_replication == 1:
            return True
        # replicated batch: identical replicas (the predict path's deterministic initial state) run like any batch; replicas that differ
        # (random initial state) couple through the termination rule -- the library has a lock-step launch for small batches of the SP
        # triple and reports anything else, which sends the batch to the step-wise loop
        self._replicas_same = self._replicas_identical(sat_problem, states)
        return self._replicas_same or model == native.MODEL_SP

    def _forward_core(self, init_propagator_state, init_decimator_state, sat_problem, iteration_num, is_training, check_termination):
        # (reference signature; `forward` passes the two states in one OwnedState box as the first argument)
        box = init_propagator_state if isinstance(init_propagator_state, OwnedState) else OwnedState((init_propagator_state, init_decimator_state))
        init_propagator_state, init_decimator_state = box[0], box[1]
        can = self._can_run_persistent(sat_problem, is_training, check_termination, tuple(init_propagator_state[:2]) + tuple(init_decimator_state[:2]))
        if self._isolated and not can:
            raise native.NativeError("isolated-instance mode runs on the persistent SP loop only (p-d-p, standard termination check)")
        if self._exchange is not None and not can and not self._neural_triple(sat_problem, is_training, check_termination):
            raise native.NativeError("a coupled forward spread over several processes runs on the persistent SP loop (p-d-p, standard termination check) "
                                     "or on a neural triple (np-nd-np / p-nd-np prediction), whose only coupling is the end of the loop")
        if can:
            out = self._forward_core_persistent(init_propagator_state, init_decimator_state, sat_problem, iteration_num, check_termination)
            if out is not None:
                return out
            if self._exchange is not None:
                raise native.CoupledForwardFailed("a coupled forward spread over several processes met a coupling only the step-wise loop "
                                                  "reproduces (batch-global minimum != 0); that loop is single-process")
        del init_propagator_state, init_decimator_state                # (see forward: this frame lets go of the initial state)
        if self._can_run_graph_loop(sat_problem, iteration_num, is_training, check_termination):
            return self._forward_core_graph(box, sat_problem, iteration_num, check_termination)
        return self._forward_core_stepwise(box, None, sat_problem, iteration_num, is_training, check_termination)

    # ---- the neural triples' loop, device-driven -------------------------------------------------------------------------------------------
    def _neural_triple(self, sat_problem, is_training, check_termination):
        """np-nd-np / p-nd-np prediction: the library's neural decimator (no variable is ever fixed: the edge mask of the first sweep is the
        mask of every sweep) behind the neural or the adaptor-fed survey propagator, the variable-side neural predictor, the trainer's own
        termination callback or none.  Such a forward couples its instances in ONE place: `active_mask.sum() <= 0` ends the loop for all of them
        (solver.py:383-384) -- which is why it can run device-driven (_forward_core_graph) and, spread over several processes, needs one bit
        per sweep from the other parts (_forward_core_stepwise with self._exchange)."""
        if is_training or (check_termination is not None and not _is_standard_termination(check_termination)):
            return False
        pr, de, pd = self._propagator, self._decimator, self._predictor
        if type(de) is not pdp_decimate.NeuralDecimator or type(pd) is not pdp_predict.NeuralPredictor:
            return False
        if not (type(pr) is pdp_propagate.NeuralMessagePasser or (type(pr) is pdp_propagate.SurveyPropagator and pr._include_adaptors)):
            return False
        if getattr(pr, '_meta_dim', 0) or de._meta_dim or pd._meta_dim or sat_problem._meta_data is not None:
            return False
        if pd._variable_classifier is None or pd._function_classifier is not None:
            return False
        return not self._isolated and not any(getattr(m, '_train_path', False) for m in (pr, de, pd))

    def _can_run_graph_loop(self, sat_problem, iteration_num, is_training, check_termination):
        "a neural triple in one process; anything else -- foreign plug-ins, training, graph features, kernel timing, a split forward -- takes the step-wise loop"
        if os.environ.get('PDP_NO_GRAPH_LOOP') or native.kernel_timing_enabled() or self._exchange is not None:
            return False
        if native.BUILD != 'parity':
            return False                                   # (the frozen fast build allocates a per-stream workspace inside its GRU call: not capturable)
        if int(iteration_num) < int(os.environ.get('PDP_GRAPH_LOOP_MIN_SWEEPS', '8')):
            return False                                   # two captures per forward: a short loop is cheaper sweep by sweep
        # Large segments gain nothing (the GPU is busy either way: configs[3]'s shard runs 6.30 against 6.34 it/s) and the second state set
        # costs memory there (117 against 91 GB reserved): the graph loop is for the small dynamic segments, where the host's round trips count
        hidden = int(getattr(self._decimator, '_hidden_dimension', 0) or 0)
        if sat_problem._edge_num * max(1, hidden) > int(float(os.environ.get('PDP_GRAPH_LOOP_MAX_STATE', '4e8'))):
            return False
        return self._neural_triple(sat_problem, is_training, check_termination)

    def _forward_core_graph(self, box, sat_problem, iteration_num, check_termination):
        """The loop of solver.py:355-386 for the neural triples without a host round trip per sweep.  The first sweep runs as in the step-wise
        loop (the library's workspaces come into being there, and the host learns whether an edge mask exists).  The body of every later sweep
        -- the same plug-in calls, writing into named buffers -- is captured ONCE per parity as a HIP graph: sweep k reads state set k - 1
        and writes set k (mod 2).  The loop's end is decided on the device (pdp_loop_step: the sweep is counted, and `active_mask.sum() <= 0`
        raises a stop word behind which the state-writing kernels of replayed sweeps return at once); the host replays graphs, looks at the
        stop word every few sweeps and takes the executed count from the device.  States, solution, mask and count equal the step-wise loop's
        (tests/test_api_forward.py::test_graph_loop_equals_the_stepwise_loop)."""
        T = int(iteration_num)
        nat = sat_problem._native
        active_mask = None if check_termination is None else torch.ones(sat_problem._batch_size, 1, dtype=torch.uint8, device=self._device)
        am_flat = None if active_mask is None else active_mask.reshape(-1)
        all_active = [True]

        def sweep(p_in, d_in, p_out, d_out, first):
            self._propagator._out, self._decimator._out = p_out, d_out
            try:
                ps = self._propagator(p_in, d_in, sat_problem, False, active_mask)
                ds = self._decimator(d_in, ps, sat_problem, False, active_mask)
            finally:
                self._propagator._out = self._decimator._out = None
            if first:
                all_active[0] = sat_problem.refresh_edge_mask()
            else:
                nat.refresh_edge_mask(False)                # same kernel, no host read of the flag (nothing is decimated: it cannot change)
            ds = tuple(ds[:2])
            if not all_active[0]:
                ds = ds + (sat_problem._edge_mask,)
            if check_termination is not None:
                prediction = self._predictor(ds, sat_problem)
                prediction = self._update_solution(prediction, sat_problem)
                check_termination(active_mask, prediction, sat_problem)
            nat.loop_step(am_flat)
            return tuple(ps[:2]), ds

        nat.loop_begin()
        ended = False
        try:
            ps_a, ds_a, ps_b, ds_b, iters = self._graph_loop_body(sweep, box, nat, T, active_mask)
            ended = True
        finally:
            if not ended:
                nat.loop_read(end=True)                     # whatever went wrong: the stop word must not outlive the loop (later calls would write nothing)
        if T > 1 and iters > 1 and iters % 2 == 0:
            ps_a, ds_a = ps_b, ds_b
        self.last_run.update(path='graph', iterations=iters)
        self._active_mask = active_mask
        return ps_a, ds_a

    def _graph_loop_body(self, sweep, box, nat, T, active_mask):
        propagator_state, decimator_state = box.take()      # the box is empty now: these two names are the only references
        ps_a, ds_a = sweep(propagator_state, decimator_state, None, None, True)
        del propagator_state, decimator_state               # the initial state goes (its four [E, H] tensors are dead after the first sweep)
        ps_b = ds_b = None
        if T > 1:
            ps_b = tuple(torch.empty_like(x) for x in ps_a[:2])
            ds_b2 = tuple(torch.empty_like(x) for x in ds_a[:2])
            ds_b = ds_b2 + tuple(ds_a[2:])
            graphs = []
            pool = None
            side = torch.cuda.Stream(device=self._device)
            side.wait_stream(torch.cuda.current_stream())
            for src_p, src_d, dst_p, dst_d in ((ps_a, ds_a, ps_b, ds_b2), (ps_b, ds_b, ps_a, tuple(ds_a[:2]))):
                g = torch.cuda.CUDAGraph()
                # (capture_begin / capture_end by hand: the torch.cuda.graph context also runs a device synchronize, gc.collect() and
                #  empty_cache() on entry -- ~1 ms per capture, two captures per segment, with the first sweep's kernels still in flight)
                with torch.cuda.stream(side):
                    g.capture_begin(*(() if pool is None else (pool,)), capture_error_mode='thread_local')
                    try:
                        sweep(src_p, src_d, dst_p, dst_d, False)
                    finally:
                        g.capture_end()
                pool = g.pool()
                graphs.append(g)
            torch.cuda.current_stream().wait_stream(side)
            look = max(1, int(os.environ.get('PDP_GRAPH_LOOP_LOOK', '8')))
            stopped = False
            for k in range(2, T + 1):
                graphs[k & 1].replay()                      # even sweeps: set a -> set b
                if active_mask is not None and (k - 1) % look == 0 and k < T:
                    stopped, iters = nat.loop_read()
                    if stopped:
                        break
        stopped, iters = nat.loop_read(end=True)
        return ps_a, ds_a, ps_b, ds_b, iters

    def _forward_core_persistent(self, init_propagator_state, init_decimator_state, sat_problem, iteration_num, check_termination):
        """the whole loop of solver.py:355-386 in one kernel launch; None if the speculation failed.  The first sweep reads the
        DECIMATOR's initial state (solver.py:365: propagator(propagator_state, decimator_state, ...)); the propagator's own initial
        state only fills instances that are inactive, and none is at the first sweep -- the two differ when the initial state is
        random (test mode, base.py:229)."""
        nat = sat_problem._native
        src = init_decimator_state if int(iteration_num) > 0 else init_propagator_state
        q = src[0].clone().contiguous()
        fs = src[1].clone().contiguous()
        active_mask = torch.ones(sat_problem._batch_size, dtype=torch.uint8, device=self._device)
        handle = self._decimator.native_handle(sat_problem)
        model = self._persistent_model()
        extra = {}
        if model == native.MODEL_REINFORCE:
            # one torch.rand(1) per executed iteration (pdp_decimate.py:218): draw them all, rewind, consume what the loop used
            rng_state = torch.get_rng_state()
            coins = torch.cat([torch.rand(1) for _ in range(int(iteration_num))]) if int(iteration_num) > 0 else torch.zeros(0)
            extra = dict(model=model, coins=coins.to(self._device), decimation_probability=self._decimator._decimation_probability)
            tolerance, t_max = 0.01, 0.0                   # the gate's constant (pdp_decimate.py:215); no counters
        else:
            tolerance, t_max = self._decimator._tolerance, self._decimator._t_max
        try:
            iters, used_lds = nat.sp_solve(q, fs, active_mask, handle, int(iteration_num), tolerance, t_max, self._propagator._pi,
                                           check_termination=check_termination is not None,
                                           replicas_identical=sat_problem._batch_replication > 1 and getattr(self, '_replicas_same', True),
                                           isolate_instances=self._isolated, inputs_disposable=True,        # q / fs are clones: init_state stays intact
                                           **extra)
        except native.SpeculationFailed:
            if model == native.MODEL_REINFORCE:
                torch.set_rng_state(rng_state)
            return None            # the library restored every array it touched; fall back to the strict step-wise loop
        if model == native.MODEL_REINFORCE:
            torch.set_rng_state(rng_state)
            for _ in range(iters):
                torch.rand(1)
        self.last_run.update(path='persistent-lds' if used_lds else 'persistent-hbm', iterations=iters)
        sat_problem._edge_mask = nat.edge_mask
        state = (q, fs)
        dec_state = state
        if iters > 0 and not bool((nat.edge_mask == 1).all().item()):
            dec_state = state + (nat.edge_mask,)
        self._active_mask = active_mask.unsqueeze(1)
        return state, dec_state

    def _forward_core_stepwise(self, init_propagator_state, init_decimator_state, sat_problem, iteration_num, is_training, check_termination):
        "generic plug-in loop, statement for statement the reference's (solver.py:355-386)"
        if isinstance(init_propagator_state, OwnedState):
            propagator_state, decimator_state = init_propagator_state.take()     # the box is empty now: these two names are the only references
        else:
            propagator_state, decimator_state = init_propagator_state, init_decimator_state
        del init_propagator_state, init_decimator_state          # one name per state: the old tensors go when a sweep replaces them
        if check_termination is None:
            active_mask = None
        else:
            active_mask = torch.ones(sat_problem._batch_size, 1, dtype=torch.uint8, device=self._device)
        iters = 0
        for _ in range(int(iteration_num)):
            propagator_state = self._propagator(propagator_state, decimator_state, sat_problem, is_training, active_mask)
            decimator_state = self._decimator(decimator_state, propagator_state, sat_problem, is_training, active_mask)
            all_active = sat_problem.refresh_edge_mask()
            if not all_active:
                decimator_state = tuple(decimator_state[:2]) + (sat_problem._edge_mask,)
            iters += 1
            if check_termination is not None:
                prediction = self._predictor(decimator_state, sat_problem)
                prediction = self._update_solution(prediction, sat_problem)
                check_termination(active_mask, prediction, sat_problem)
                still = int(active_mask.sum().item()) > 0
                if self._exchange is not None:
                    # a part of a coupled forward: the loop ends when no instance of ANY part is active (one bit, OR over the parts)
                    word = np.array([1 if still else 0], dtype=np.uint32)
                    self._exchange(np.zeros(0, np.uint32), np.zeros(0, np.uint32), word)
                    still = bool(word[0])
                if not still:
                    break
        self.last_run.update(path='stepwise', iterations=iters)
        self._active_mask = active_mask
        return propagator_state, decimator_state

    # -----------------------------------------------------------------------------------------------------------
    def _update_solution(self, prediction, sat_problem):
        "solver.py:388-399"
        if prediction[0] is not None and torch.is_grad_enabled() and prediction[0].requires_grad:
            # training: the blend stays on the autograd graph (every variable is active there: no simplification, solver.py:332)
            av = sat_problem._active_variables
            variable_solution = av * prediction[0] + (1.0 - av) * sat_problem._solution.unsqueeze(1)
            return variable_solution, prediction[1]
        if prediction[0] is not None:
            variable_solution = sat_problem._native.update_solution(prediction[0].reshape(-1).contiguous())
        else:
            variable_solution = None
        return variable_solution, prediction[1]

    def _deduplicate(self, prediction, propagator_state, decimator_state, sat_problem):
        "min-energy replica per original instance (solver.py:401-431 with the integer-division fix)"
        if sat_problem._batch_replication <= 1:
            return None, None, None
        R = sat_problem._batch_replication
        variable_prediction, chosen = sat_problem._native.deduplicate(prediction[0].reshape(-1).contiguous())
        e0 = sat_problem._edge_num // R
        b0 = sat_problem._batch_size // R
        # per-edge replica choice: instance of each original edge
        inst_of_edge = sat_problem._orig[1][sat_problem._orig[0][0].long()].long()
        pick = chosen.long()[inst_of_edge]                       # [E0]
        idx = pick * e0 + torch.arange(e0, device=self._device)
        sel = lambda states: None if states is None else tuple(x[idx] for x in states)
        return (variable_prediction, None), sel(propagator_state), sel(decimator_state)

    def _local_search(self, prediction, sat_problem, batch_replication):
        "Walk-SAT post-processing (solver.py:433-467)"
        pred = prediction[0].reshape(-1).contiguous()
        w = int(self._local_search_iterations)
        nat = sat_problem._native
        if self._rng == 'torch' and w > 0:
            # The reference draws rand(V, 1) and rand(B) from the global CPU generator in every step it executes (solver.py:457,460).
            # The stream is drawn in pieces of whole steps (bounded host / device memory, nothing drawn past an early stop by more than
            # one piece): a piece is one call of the native search, which resumes from the assignment the previous piece left -- the
            # search state IS the assignment ((a + 1) / 2 maps back to a for active variables, inactive ones stay 0).
            V, B = sat_problem._variable_num, sat_problem._batch_size
            per_piece = max(1, int(os.environ.get('PDP_WALKSAT_RNG_CHUNK', str(1 << 26))) // (V + B))
            out, steps = pred, 0
            while steps < w:
                c = min(per_piece, w - steps)
                state = torch.get_rng_state()
                draws = torch.rand(c * (V + B)).view(c, V + B)
                var_rand = draws[:, :V].contiguous().to(self._device)
                coin = draws[:, V:].contiguous().to(self._device)
                out, done = nat.local_search(out.reshape(-1).contiguous(), c, self._epsilon, var_rand, coin)
                steps += done
                if done < c:
                    torch.set_rng_state(state)
                    if done > 0:
                        torch.rand(done * (V + B))     # consume exactly what the reference consumes
                    break
        else:
            try:
                out, steps = nat.local_search(pred, w, self._epsilon, seed=self._seed)
            except native.SpeculationFailed as ex:
                if self._exchange is None:
                    raise
                # one part of a coupled forward cannot take the persistent search, or a step's batch-global minimum was not 0 anywhere: every
                # part gets this status (the library agrees on it across the parts), and the segment is solved whole by one process
                raise native.CoupledForwardFailed(str(ex))
        sat_problem._edge_mask = nat.edge_mask
        self.last_run['walksat_steps'] = steps
        return out, prediction[1]

    def get_init_state(self, graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, randomized, batch_replication=1):
        "solver.py:498-511"
        if self._propagator is None:
            init_propagator_state = None
        else:
            init_propagator_state = self._propagator.get_init_state(graph_map, batch_variable_map, batch_function_map,
                                                                    edge_feature, graph_feat, randomized, batch_replication)
        if self._decimator is None:
            init_decimator_state = None
        else:
            init_decimator_state = self._decimator.get_init_state(graph_map, batch_variable_map, batch_function_map,
                                                                  edge_feature, graph_feat, randomized, batch_replication)
        return init_propagator_state, init_decimator_state


###############################################################


class NeuralPropagatorDecimatorSolver(PropagatorDecimatorSolverBase):
    "Fully neural PDP solver: neural propagator + neural (GRU) decimator + neural predictor (reference: solver.py:517-537)."

    def __init__(self, device, name, edge_dimension, meta_data_dimension, propagator_dimension, decimator_dimension,
                 mem_hidden_dimension, agg_hidden_dimension, mem_agg_hidden_dimension, prediction_dimension,
                 variable_classifier=None, function_classifier=None, dropout=0, local_search_iterations=0, epsilon=0.05,
                 rng='torch', seed=0):
        super(NeuralPropagatorDecimatorSolver, self).__init__(
            device=device, name=name,
            propagator=pdp_propagate.NeuralMessagePasser(device, edge_dimension, decimator_dimension, meta_data_dimension,
                                                         propagator_dimension, mem_hidden_dimension, mem_agg_hidden_dimension,
                                                         agg_hidden_dimension, dropout),
            decimator=pdp_decimate.NeuralDecimator(device, propagator_dimension, meta_data_dimension, decimator_dimension,
                                                   mem_hidden_dimension, mem_agg_hidden_dimension, agg_hidden_dimension,
                                                   edge_dimension, dropout),
            predictor=pdp_predict.NeuralPredictor(device, decimator_dimension, prediction_dimension, edge_dimension,
                                                  meta_data_dimension, mem_hidden_dimension, agg_hidden_dimension,
                                                  mem_agg_hidden_dimension, variable_classifier, function_classifier),
            local_search_iterations=local_search_iterations, epsilon=epsilon, rng=rng, seed=seed)


class NeuralSurveyPropagatorSolver(PropagatorDecimatorSolverBase):
    """SP propagator with learned adaptors + neural decimator + neural predictor (reference: solver.py:543-561, model type p-nd-np).
    The reference builds its decimator with a function-message width of 1 against a propagator that emits [eta, force], and fails in
    the first GRU call; here the width is the 2 the propagator has -- the golden trace this model is pinned to
    (tests/golden/trace_p_nd_np.npz) comes from the reference with exactly that one-word change."""

    def __init__(self, device, name, edge_dimension, meta_data_dimension, decimator_dimension, mem_hidden_dimension,
                 agg_hidden_dimension, mem_agg_hidden_dimension, prediction_dimension, variable_classifier=None,
                 function_classifier=None, dropout=0, local_search_iterations=0, epsilon=0.05, rng='torch', seed=0):
        super(NeuralSurveyPropagatorSolver, self).__init__(
            device=device, name=name,
            propagator=pdp_propagate.SurveyPropagator(device, decimator_dimension, include_adaptors=True),
            decimator=pdp_decimate.NeuralDecimator(device, (3, 2), meta_data_dimension, decimator_dimension, mem_hidden_dimension,
                                                   mem_agg_hidden_dimension, agg_hidden_dimension, edge_dimension, dropout),
            predictor=pdp_predict.NeuralPredictor(device, decimator_dimension, prediction_dimension, edge_dimension,
                                                  meta_data_dimension, mem_hidden_dimension, agg_hidden_dimension,
                                                  mem_agg_hidden_dimension, variable_classifier, function_classifier),
            local_search_iterations=local_search_iterations, epsilon=epsilon, rng=rng, seed=seed)


class NeuralSequentialDecimatorSolver(PropagatorDecimatorSolverBase):
    """Neural propagator + the sequential decimator scored by a neural predictor + identity predictor
    (reference: solver.py:616-637, model type np-d-np).  The decimator's survey gate reads column 0 of the neural
    function state, a logsigmoid output, so under ``check_termination`` the reference switches every instance off after
    the first iteration; that behaviour is reproduced, not repaired."""

    def __init__(self, device, name, edge_dimension, meta_data_dimension, propagator_dimension, decimator_dimension,
                 mem_hidden_dimension, agg_hidden_dimension, mem_agg_hidden_dimension, classifier_dimension, dropout,
                 tolerance, t_max, local_search_iterations=0, epsilon=0.05, rng='torch', seed=0):
        super(NeuralSequentialDecimatorSolver, self).__init__(
            device=device, name=name,
            propagator=pdp_propagate.NeuralMessagePasser(device, edge_dimension, decimator_dimension, meta_data_dimension,
                                                         propagator_dimension, mem_hidden_dimension, mem_agg_hidden_dimension,
                                                         agg_hidden_dimension, dropout),
            decimator=pdp_decimate.SequentialDecimator(
                device, message_dimension=(3, 1),
                scorer=pdp_predict.NeuralPredictor(device, decimator_dimension, 1, edge_dimension, meta_data_dimension,
                                                   mem_hidden_dimension, agg_hidden_dimension, mem_agg_hidden_dimension,
                                                   variable_classifier=util.PerceptronTanh(decimator_dimension, classifier_dimension, 1),
                                                   function_classifier=None),
                tolerance=tolerance, t_max=t_max),
            predictor=pdp_predict.IdentityPredictor(device=device, random_fill=True, rng=rng, seed=seed),
            local_search_iterations=local_search_iterations, epsilon=epsilon, rng=rng, seed=seed)


def build_neural_solver(device, config, perceptron_cls, common):
    "model types np-nd-np, p-nd-np and np-d-np (reference: trainer.py:51-81)"
    if config['model_type'] == 'np-d-np':
        return NeuralSequentialDecimatorSolver(
            device=device, name=config['model_name'], edge_dimension=config['edge_feature_dim'],
            meta_data_dimension=config['meta_feature_dim'], propagator_dimension=config['hidden_dim'],
            decimator_dimension=config['hidden_dim'], mem_hidden_dimension=config['mem_hidden_dim'],
            agg_hidden_dimension=config['agg_hidden_dim'], mem_agg_hidden_dimension=config['mem_agg_hidden_dim'],
            classifier_dimension=config['classifier_dim'], dropout=config.get('dropout', 0),
            tolerance=config['tolerance'], t_max=config['t_max'], **common)
    if config['model_type'] == 'p-nd-np':
        return NeuralSurveyPropagatorSolver(
            device=device, name=config['model_name'], edge_dimension=config['edge_feature_dim'],
            meta_data_dimension=config['meta_feature_dim'], decimator_dimension=config['hidden_dim'],
            mem_hidden_dimension=config['mem_hidden_dim'], agg_hidden_dimension=config['agg_hidden_dim'],
            mem_agg_hidden_dimension=config['mem_agg_hidden_dim'], prediction_dimension=config['prediction_dim'],
            variable_classifier=perceptron_cls(config['hidden_dim'], config['classifier_dim'], config['prediction_dim']),
            function_classifier=None, dropout=config.get('dropout', 0), **common)
    if config['model_type'] != 'np-nd-np':
        raise NotImplementedError("model_type %r has no native implementation (np-nd-np, np-d-np, p-nd-np, p-d-p, walk-sat, reinforce are available)"
                                  % (config['model_type'],))
    return NeuralPropagatorDecimatorSolver(
        device=device, name=config['model_name'], edge_dimension=config['edge_feature_dim'],
        meta_data_dimension=config['meta_feature_dim'], propagator_dimension=config['hidden_dim'],
        decimator_dimension=config['hidden_dim'], mem_hidden_dimension=config['mem_hidden_dim'],
        agg_hidden_dimension=config['agg_hidden_dim'], mem_agg_hidden_dimension=config['mem_agg_hidden_dim'],
        prediction_dimension=config['prediction_dim'],
        variable_classifier=perceptron_cls(config['hidden_dim'], config['classifier_dim'], config['prediction_dim']),
        function_classifier=None, dropout=config.get('dropout', 0), **common)


class SurveyPropagatorSolver(PropagatorDecimatorSolverBase):
    "Classical SP-guided decimation via the PDP framework (reference: solver.py:567-578)."

    def __init__(self, device, name, tolerance, t_max, local_search_iterations=0, epsilon=0.05, rng='torch', seed=0, persistent=True):
        super(SurveyPropagatorSolver, self).__init__(
            device=device, name=name,
            propagator=pdp_propagate.SurveyPropagator(device, decimator_dimension=1, include_adaptors=False),
            decimator=pdp_decimate.SequentialDecimator(
                device, message_dimension=(3, 1),
                scorer=pdp_predict.SurveyScorer(device, message_dimension=1, include_adaptors=False),
                tolerance=tolerance, t_max=t_max),
            predictor=pdp_predict.IdentityPredictor(device=device, random_fill=True, rng=rng, seed=seed),
            local_search_iterations=local_search_iterations, epsilon=epsilon, rng=rng, seed=seed, persistent=persistent)


class WalkSATSolver(PropagatorDecimatorSolverBase):
    "Classical Walk-SAT via the PDP framework (reference: solver.py:584-592)."

    def __init__(self, device, name, iteration_num, epsilon=0.05, rng='torch', seed=0):
        super(WalkSATSolver, self).__init__(
            device=device, name=name, propagator=None, decimator=None,
            predictor=pdp_predict.IdentityPredictor(device=device, random_fill=True, rng=rng, seed=seed),
            local_search_iterations=iteration_num, epsilon=epsilon, rng=rng, seed=seed)


class ReinforceSurveyPropagatorSolver(PropagatorDecimatorSolverBase):
    "Classical Reinforce via the PDP framework (reference: solver.py:598-610)."

    def __init__(self, device, name, pi=0.1, decimation_probability=0.5, local_search_iterations=0, epsilon=0.05, rng='torch', seed=0,
                 persistent=True):
        super(ReinforceSurveyPropagatorSolver, self).__init__(
            device=device, name=name,
            propagator=pdp_propagate.SurveyPropagator(device, decimator_dimension=1, include_adaptors=False, pi=pi),
            decimator=pdp_decimate.ReinforceDecimator(
                device, scorer=pdp_predict.SurveyScorer(device, message_dimension=1, include_adaptors=False, pi=pi),
                decimation_probability=decimation_probability),
            predictor=pdp_predict.ReinforcePredictor(device=device),
            local_search_iterations=local_search_iterations, epsilon=epsilon, rng=rng, seed=seed, persistent=persistent)

"""Predict and test drivers (reference: src/pdp/factorgraph/base.py, predict path :252-305,451-472, test path :183-250,406-449).

``train`` / ``_train_epoch`` / ``_train_batch`` (reference: base.py:113-182, 311-404) run the neural solver's differentiable path
(pdp/nn/train_ops.py: native forward / adjoint pairs under torch autograd) with the optimizer the caller provides; ``predict`` and
``test`` keep their signatures.  There is no ``nn.DataParallel`` wrap (it mis-scatters ``graph_map``, SURVEY.md
App. B-13): multi-GPU runs shard instances across ranks instead (pdp/parallel.py).
"""

import os
import time

import numpy as np
import torch

from pdp import native
from pdp.factorgraph.dataset import FactorGraphDataset
from pdp.nn.solver import OwnedState


class FactorGraphTrainerBase(object):
    "Base class of the prediction pipeline (abstract)."

    def __init__(self, config, has_meta_data, error_dim, loss, evaluator, use_cuda, logger):
        self._config = config
        self._logger = logger
        if not use_cuda:
            raise native.NativeError("cpu_mode is not available: this build runs the PDP hot path on the MI355X only "
                                     "(the CPU restatement under oracle/ is test infrastructure, not a fallback)")
        native.require_gpu()
        self._use_cuda = True
        self._device = torch.device('cuda', torch.cuda.current_device())
        self._error_dim = error_dim
        self._loss = loss
        self._evaluator = evaluator
        if config.get('verbose'):
            self._logger.info('Using GPU %s...' % torch.cuda.get_device_name(self._device))
        self._model_list = [m.to(self._device) for m in self._build_graph(self._config)]

    def _build_graph(self, config):
        raise NotImplementedError("Subclass must implement abstract method")

    def _load(self, import_path_base):
        for model in self._model_list:
            model.load(import_path_base)

    def _save(self, export_path_base):
        for model in self._model_list:
            model.save(export_path_base)

    def _to_cuda(self, data):
        if isinstance(data, list) or data is None:
            return data
        return data.to(self._device, non_blocking=True)

    # ---- training (reference: base.py:108-182, 311-404) -------------------------------------------------------------------------
    def get_parameter_list(self):
        "list of dictionaries with the models' trainable parameters, for the optimizer's constructor (base.py:108-111)"
        return [{'params': [p for p in model.parameters() if p.requires_grad]} for model in self._model_list]

    def _reset_global_step(self):
        for model in self._model_list:
            model._global_step.data.zero_()

    def _compute_loss(self, model, loss, prediction, label, graph_map, batch_variable_map, batch_function_map, edge_feature, meta_data):
        return loss(prediction, label)

    def _train_batch(self, total_loss, optimizer, graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, label):
        """One optimizer step on one (segment of a) batch (reference: base.py:149-182): train_outer_recurrence_num calls of the model, the state
        carried from call to call, loss = sum_t lambda^(T - t - 1) loss_t, back-propagation through all of them, gradient clipping per model."""
        import torch.nn as nn
        optimizer.zero_grad()
        T = int(self._config['train_outer_recurrence_num'])
        lam = float(self._config['lambda'])
        for (i, model) in enumerate(self._model_list):
            state = model.get_init_state(graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, self._config['randomized'])
            loss = torch.zeros(1, device=self._device)
            for t in range(T):
                prediction, state = model(init_state=state, graph_map=graph_map, batch_variable_map=batch_variable_map,
                                          batch_function_map=batch_function_map, edge_feature=edge_feature, meta_data=graph_feat,
                                          is_training=True, iteration_num=self._config['train_inner_recurrence_num'])
                loss = loss + self._compute_loss(model=model, loss=self._loss, prediction=prediction, label=label, graph_map=graph_map,
                                                 batch_variable_map=batch_variable_map, batch_function_map=batch_function_map,
                                                 edge_feature=edge_feature, meta_data=graph_feat) * (lam ** (T - t - 1))
            loss.backward()
            nn.utils.clip_grad_norm_(model.parameters(), self._config['clip_norm'])
            total_loss[i] += float(loss.detach().cpu().numpy().reshape(-1)[0])
        optimizer.step()

    def _train_epoch(self, train_loader, optimizer):
        "reference: base.py:113-147 -- the models' global step advances once per loader batch; returns the loss per example"
        total_loss = np.zeros(len(self._model_list), dtype=np.float32)
        total_example_num = 0
        for data in train_loader:
            for i in range(len(data[0])):
                (graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, label, _) = [self._to_cuda(d[i]) for d in data]
                total_example_num += int(batch_variable_map.max().item()) + 1
                self._train_batch(total_loss, optimizer, graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, label)
            for model in self._model_list:
                model._global_step.data += 1
        return total_loss / max(1, total_example_num)

    # model types whose training the reference itself cannot run (tests/golden/train_np_d_np_reference.json records its exception): np-d-np's
    # prediction is sat_problem._solution -- no parameter reaches it except through values a decimation wrote -- and set_variables edits, in
    # place, the flag tensors the scorer's autograd graph saved, so loss.backward() raises.  The classical types have no parameters.
    UNTRAINABLE = {'np-d-np': "the reference's loss.backward() raises for this model type (identity predictor over the in-place edited solution; "
                              "tests/golden/train_np_d_np_reference.json)",
                   'p-d-p': 'no trainable parameters', 'walk-sat': 'no trainable parameters', 'reinforce': 'no trainable parameters'}

    def _loader(self, path, limit_key, shuffle, generator=None, epoch_size=0, batch_replication=1):
        return FactorGraphDataset.get_loader(
            input_file=path, limit=self._config[limit_key], hidden_dim=self._config['hidden_dim'], batch_size=self._config['batch_size'],
            shuffle=shuffle, num_workers=0, max_cache_size=self._config.get('max_cache_size', 100000), generator=generator,
            epoch_size=epoch_size, batch_replication=batch_replication)

    def _restore(self, which, last_dir, best_dir):
        "load_model = 'best' | 'last': start a repetition from that checkpoint when its directory was given"
        chosen = {'best': best_dir, 'last': last_dir}.get(which)
        if chosen is not None:
            self._load(chosen)

    def _report_epoch(self, rep, epoch, errors, losses, seconds):
        parts = ['Step {:d}: {:s} error={:s}, {:s} loss={:5.5f} |'.format(int(m._global_step.int()[0]), m._name,
                                                                          np.array_str(errors[:, i].flatten()), m._name, losses[i])
                 for i, m in enumerate(self._model_list)]
        self._logger.info('Rep {:2d}, Epoch {:2d}: {:s}'.format(rep + 1, epoch + 1, ''.join(parts)))
        self._logger.info('Time spent: %s seconds' % seconds)

    def train(self, train_list, validation_list, optimizer, last_export_path_base=None, best_export_path_base=None, metric_index=0,
              load_model=None, reset_step=False, generator=None, train_epoch_size=0):
        """Trains the models (contract of the reference's train(), base.py:311-404): ``repetition_num`` repetitions of ``epoch_num`` epochs;
        an epoch is one pass over the training set (shuffled file, or ``train_epoch_size`` generated instances) followed by a validation
        pass with the test-mode metrics.  After every epoch the models go to ``last_export_path_base``; a model goes to
        ``best_export_path_base`` whenever its validation metric ``metric_index`` improves; at the end the loss / error histories are
        stored there as losses.npy / errors.npy.  Returns (models, errors [error_dim, models, epochs, repetitions], losses [models,
        epochs, repetitions])."""
        why = self.UNTRAINABLE.get(self._config.get('model_type'))
        if why is not None:
            raise native.NativeError("model_type %r cannot be trained: %s" % (self._config.get('model_type'), why))
        cfg = self._config
        n_models, n_epochs, n_reps = len(self._model_list), cfg['epoch_num'], cfg['repetition_num']
        feed = self._loader(train_list[0], 'train_batch_limit', True, generator, train_epoch_size)
        held_out = self._loader(validation_list[0], 'test_batch_limit', False)
        errors = np.zeros((self._error_dim, n_models, n_epochs, n_reps), dtype=np.float32)
        losses = np.zeros((n_models, n_epochs, n_reps), dtype=np.float32)
        best_so_far = np.full(n_models, np.inf)
        for rep in range(n_reps):
            self._restore(load_model, last_export_path_base, best_export_path_base)
            if reset_step:
                self._reset_global_step()
            for epoch in range(n_epochs):
                t0 = time.time()
                losses[:, epoch, rep] = self._train_epoch(feed, optimizer)
                errors[:, :, epoch, rep] = self._test_epoch(held_out, 1)
                torch.cuda.synchronize()
                seconds = time.time() - t0
                if last_export_path_base is not None:
                    self._save(last_export_path_base)
                if best_export_path_base is not None:
                    score = errors[metric_index, :, epoch, rep]
                    for i in np.nonzero(score < best_so_far)[0]:
                        best_so_far[i] = score[i]
                        self._model_list[i].save(best_export_path_base)
                if cfg.get('verbose'):
                    self._report_epoch(rep, epoch, errors[:, :, epoch, rep], losses[:, epoch, rep], seconds)
        if best_export_path_base is not None:
            where = os.path.relpath(best_export_path_base)
            for name, history in (('losses', losses), ('errors', errors)):
                np.save(os.path.join(where, name), history, allow_pickle=False)
            self._save(best_export_path_base)
        return self._model_list, errors, losses

    def _predict_epoch(self, validation_loader, post_processor, batch_replication, file, units=None):
        """reference: base.py:252-278.  Every forward is keyed by the GLOBAL (loader batch, segment) index of its unit: the loader says where
        a batch sits in the run (``LoaderBatch.index`` / ``.segments``); a loader without that information (torch's DataLoader, a wrapper)
        yields every batch in order, so the position in the iteration is the index -- it cannot be a sharded one.  With ``units`` (a list)
        the rows of a unit are appended to it as ((batch, segment), text) instead of being written to ``file``."""
        import io
        from pdp import parallel
        base_seed = int(self._config.get('random_seed', 0) or 0)
        with torch.no_grad():
            for position, data in enumerate(validation_loader):
                j = getattr(data, 'index', None)
                segment_ids = getattr(data, 'segments', None)
                if j is None or segment_ids is None:
                    if units is not None and self._world() > 1:
                        raise native.NativeError("a run on several ranks needs the loader to say which (batch, segment) units it yields "
                                                 "(pdp.factorgraph.dataset.LoaderBatch); this loader does not")
                    j, segment_ids = position, list(range(len(data[0])))
                parts = getattr(data, 'parts', None)         # isolated instances dealt to ranks: (part, first variable, first instance)
                for k in range(len(data[0])):
                    for model in self._model_list:
                        if hasattr(model, 'set_random_key'):
                            model.set_random_key(parallel.batch_seed(base_seed, j, segment_ids[k]), *(parts[k][1:] if parts else ()))
                    (graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, label, misc_data) = \
                        [self._to_cuda(d[k]) for d in data]
                    sink = io.StringIO() if units is not None else file
                    stats_before = list(getattr(self, '_run_stats', []))
                    try:
                        self._predict_batch(graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat,
                                            label, misc_data, post_processor, batch_replication, sink)
                    except native.CoupledForwardFailed:
                        # --split-forward, and this segment's speculation failed (every part raises: the outcome is agreed on across the parts
                        # before any later collective).  Small batches do that -- the exact zero the persistent solver counts on is supplied by
                        # SOME variable of a large batch.  The rank of the first part solves the segment whole, with the single-process loops
                        # (lock-step / step-wise) and the segment's own Philox key; the other parts contribute no rows for this unit.
                        whole = getattr(data, 'whole', None)
                        if whole is None:
                            raise
                        if self._config.get('verbose'):
                            self._logger.info('segment (%d, %d): the coupled forward needs the single-process loop; solved whole on rank of part 0' % (j, segment_ids[k]))
                        sink = io.StringIO() if units is not None else file
                        if stats_before:
                            self._run_stats[:] = stats_before      # (models of the list that finished before the failing one counted their rows already)
                        if parts[k][0] == 0:
                            seg = whole[k]()
                            saved = [(m, getattr(m, '_exchange', None)) for m in self._model_list]
                            try:
                                for m in self._model_list:
                                    if hasattr(m, '_exchange'):
                                        m._exchange = None
                                    if hasattr(m, 'set_random_key'):
                                        m.set_random_key(parallel.batch_seed(base_seed, j, segment_ids[k]))
                                t_ = lambda a: self._to_cuda(torch.from_numpy(a))
                                self._predict_batch(t_(seg['graph_map']), t_(seg['batch_variable_map']), t_(seg['batch_function_map']), t_(seg['edge_feature']), None,
                                                    t_(seg['label']), seg['misc_data'], post_processor, batch_replication, sink)
                            finally:
                                for m, ex in saved:
                                    if hasattr(m, '_exchange'):
                                        m._exchange = ex
                    if units is not None:
                        units.append(((int(j), int(segment_ids[k])) + ((int(parts[k][0]),) if parts else ()), sink.getvalue()))

    @staticmethod
    def _world():
        return torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1

    def _draws_from_the_host_stream(self, model):
        "does a predict call of this model consume the global torch CPU generator (random fill, Walk-SAT coins, the Reinforce coin)?"
        if int(getattr(model, '_local_search_iterations', 0) or 0) > 0:
            return True
        predictor, decimator = getattr(model, '_predictor', None), getattr(model, '_decimator', None)
        return bool(getattr(predictor, '_random_fill', False)) or type(decimator).__name__ == 'ReinforceDecimator'

    def _predict_batch(self, graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, label, misc_data,
                       post_processor, batch_replication, file):
        "reference: base.py:280-305"
        for model in self._model_list:
            # the initial state is handed over, not kept (and `forward` is called without nn.Module's wrapper frames, which would hold the
            # keyword arguments for the whole call): its [E, H] tensors are released after the first sweep (pdp.nn.solver: forward)
            prediction, _ = model.forward(init_state=OwnedState(model.get_init_state(graph_map, batch_variable_map, batch_function_map, edge_feature,
                                                                                     graph_feat, randomized=False, batch_replication=batch_replication)),
                                          graph_map=graph_map, batch_variable_map=batch_variable_map,
                                          batch_function_map=batch_function_map, edge_feature=edge_feature, meta_data=graph_feat,
                                          is_training=False, iteration_num=self._config['test_recurrence_num'],
                                          check_termination=self._check_recurrence_termination, batch_replication=batch_replication)
            if post_processor is not None and callable(post_processor):
                message = post_processor(model, prediction, graph_map, batch_variable_map, batch_function_map,
                                         edge_feature, graph_feat, label, misc_data)
                print(message, file=file)

    def _check_recurrence_termination(self, active, prediction, sat_problem):
        pass

    def _compute_evaluation_metrics(self, model, evaluator, prediction, label, graph_map, batch_variable_map, batch_function_map,
                                    edge_feature, meta_data):
        return evaluator(prediction, label)

    def _test_epoch(self, validation_loader, batch_replication):
        "reference: base.py:183-219 -- per-example weighted mean of the evaluation metrics over the loader"
        with torch.no_grad():
            error = np.zeros((self._error_dim, len(self._model_list)), dtype=np.float32)
            total_example_num = 0
            for data in validation_loader:
                for i in range(len(data[0])):
                    (graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, label, _) = \
                        [self._to_cuda(d[i]) for d in data]
                    total_example_num += int(batch_variable_map.max().item()) + 1
                    self._test_batch(error, graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, label,
                                     batch_replication)
        self._last_test_counts = (error.copy(), total_example_num)       # sums, for a sharded run's single all-reduce
        return error / total_example_num

    def _test_batch(self, error, graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, label, batch_replication):
        "reference: base.py:221-250 (random initial state, like the reference's validation / test passes)"
        this_batch_size = float(int(batch_variable_map.max().item()) + 1)
        for (i, model) in enumerate(self._model_list):
            state = model.get_init_state(graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat,
                                         randomized=True, batch_replication=batch_replication)
            prediction, _ = model(init_state=state, graph_map=graph_map, batch_variable_map=batch_variable_map,
                                  batch_function_map=batch_function_map, edge_feature=edge_feature, meta_data=graph_feat,
                                  is_training=False, iteration_num=self._config['test_recurrence_num'],
                                  check_termination=self._check_recurrence_termination, batch_replication=batch_replication)
            error[:, i] += (this_batch_size * self._compute_evaluation_metrics(
                model=model, evaluator=self._evaluator, prediction=prediction, label=label, graph_map=graph_map,
                batch_variable_map=batch_variable_map, batch_function_map=batch_function_map, edge_feature=edge_feature,
                meta_data=graph_feat)).detach().cpu().numpy()

    @staticmethod
    def _test_inputs(test_list):
        "a list of files as given, the .json files of a directory, or one file; None for anything else"
        if isinstance(test_list, list):
            return test_list
        if isinstance(test_list, str) and os.path.isdir(test_list):
            inside = (os.path.join(test_list, name) for name in os.listdir(test_list))
            return [f for f in inside if os.path.isfile(f) and f.lower().endswith('.json')]
        return [test_list] if isinstance(test_list, str) else None

    def test(self, test_list, import_path_base=None, batch_replication=1):
        """Test-mode metrics per input file (contract of the reference's test(), base.py:406-449): returns
        [[file, errors [error_dim, models], seconds], ...]; the checkpoint under ``import_path_base`` is loaded before every file."""
        files = self._test_inputs(test_list)
        if files is None:
            return None
        rows = []
        for path in files:
            loader = self._loader(path, 'test_batch_limit', False, batch_replication=batch_replication)
            if import_path_base is not None:
                self._load(import_path_base)
            t0 = time.time()
            error = self._test_epoch(loader, batch_replication)
            torch.cuda.synchronize()
            seconds = time.time() - t0
            if self._config.get('verbose'):
                self._logger.info(''.join('{:s}, dataset:{:s} error={:s}|'.format(m._name, path, np.array_str(error[:, i].flatten()))
                                          for i, m in enumerate(self._model_list)))
                self._logger.info('Time spent: %s seconds' % seconds)
            rows.append([path, error, seconds])
        return rows

    def predict(self, test_list, out_file, import_path_base=None, post_processor=None, batch_replication=1):
        """Produces predictions for a (trained) PDP model (reference: base.py:451-472).

        Under ``torch.distributed`` (one process per GPU, ``python -m torch.distributed.run --nproc-per-node N satyr.py ...``) every rank
        forms the loader batches of the single-process run, cuts them into the same dynamic segments and solves the segments dealt to it
        on its own GPU -- no collective on the data path -- and the ranks meet once: an all-reduce(sum) of [instances, solved, unsat
        clauses] and a gather of the result rows by unit index, which rank 0 writes in single-process order.  One forward = one segment
        is the reference's coupling domain (batch-global minima, NaN poisoning) and the random numbers are keyed by the global (batch,
        segment) index, so the rows are those of the single-process run whatever the rank count.  That needs the counter-based
        generator: the reference's one sequential CPU stream (``rng='torch'``) is consumed unit after unit in data-dependent amounts and
        cannot be split.  A model that draws nothing from it (np-nd-np / p-nd-np without Walk-SAT) runs on N ranks as it is; one that
        does is switched to ``philox`` with a warning -- its rows then equal the single-process ``--rng philox`` rows."""
        import io
        from pdp import parallel
        distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
        world = torch.distributed.get_world_size() if distributed else 1
        rank = torch.distributed.get_rank() if distributed else 0
        rng_saved = []                                   # (object, its _rng before this call): the switch below lasts for this call only
        if world > 1 and self._config.get('rng', 'torch') != 'philox':
            for model in self._model_list:
                if self._draws_from_the_host_stream(model):
                    rng_saved.append((model, model._rng))
                    if hasattr(model._predictor, '_rng'):
                        rng_saved.append((model._predictor, model._predictor._rng))
                    self._logger.warning("%d ranks: model %s draws random numbers; the reference's sequential CPU stream (--rng torch) cannot be "
                                         "dealt to ranks, switching to --rng philox (the rows equal a single-process --rng philox run)"
                                         % (world, model._name))
                    model._rng = 'philox'
                    if hasattr(model._predictor, '_rng'):
                        model._predictor._rng = 'philox'
        # --isolated removes the couplings inside a forward: there the unit that may move is the instance, and every segment is cut into one
        # contiguous instance range per rank (a part; its Philox counters start where the part starts, SATProblem / pdp_problem_set_rng_base)
        isolated = all(getattr(m, '_isolated', False) for m in self._model_list)
        coupled = bool(self._config.get('split_forward')) and not isolated
        if coupled and world > 1 and int(batch_replication) != 1:
            raise native.NativeError("--split-forward: without batch replication (replica r of variable v has index v + r V: no contiguous parts)")
        # (PDP_DIST_FORCE=1: also in a group of ONE rank -- the exchange's collective then runs through RCCL on a one-GPU box)
        forced = distributed and os.environ.get('PDP_DIST_FORCE') == '1'
        split = (world > 1 or forced) and int(batch_replication) == 1 and (isolated or coupled)
        if split and self._config.get('verbose'):
            self._logger.info('%s: every segment is cut into %d instance ranges, one per rank'
                              % ('isolated instances' if isolated else 'coupled forwards spread over the ranks (--split-forward)', world))
        for model in self._model_list:
            if hasattr(model, '_exchange'):
                # --split-forward: the reference's batch-wide reductions are completed across the parts (pdp/parallel.py: make_exchange)
                model._exchange = parallel.make_exchange(self._device) if (split and coupled) else None
        test_loader = FactorGraphDataset.get_loader(
            input_file=test_list, limit=self._config['test_batch_limit'], hidden_dim=self._config['hidden_dim'],
            batch_size=self._config['batch_size'], shuffle=False, num_workers=0,
            max_cache_size=self._config.get('max_cache_size', 100000), batch_replication=batch_replication,
            shard=(rank, world) if (world > 1 or split) else None, split_instances=split, split_coupled=split and coupled)
        if import_path_base is not None:
            self._load(import_path_base)
        start_time = time.time()
        self._run_stats = [0, 0, 0]                       # instances, solved, unsatisfied clauses (filled by the post-processor)
        units = [] if distributed else None
        try:
            self._predict_epoch(test_loader, post_processor, batch_replication, out_file, units=units)
        finally:
            for obj, rng in rng_saved:                    # a later single-process call on the same trainer draws from the host stream again
                obj._rng = rng
        torch.cuda.synchronize()
        if distributed:
            # the one collective of the path (RCCL when the group's backend is nccl: the counters live on the GPU then), also at world size 1
            on_gpu = torch.distributed.get_backend() == 'nccl'
            self.last_stats = parallel.reduce_stats(*self._run_stats, device=self._device if on_gpu else None)
            self.last_stats['ranks'] = world
            self.last_stats['backend'] = torch.distributed.get_backend()
            if self._config.get('verbose'):
                self._logger.info('rank %d of %d solved %d units (forward calls): %s' % (rank, world, len(units), [u for u, _ in units][:32]))
            t_gather = time.time()
            parts = parallel.gather_units(units)
            t_gather = time.time() - t_gather
            if rank == 0:
                out_file.write("".join(parts))
            if self._config.get('verbose'):
                ex = [m._exchange for m in self._model_list if getattr(m, '_exchange', None) is not None]
                self._logger.info('rank %d of %d: gather of the result rows %.1f ms; %d exchanges of the coupled forwards, %.3f ms each'
                                  % (rank, world, 1e3 * t_gather, sum(e.calls for e in ex), 1e3 * sum(e.seconds for e in ex) / max(1, sum(e.calls for e in ex))))
        else:
            self.last_stats = parallel.reduce_stats(*self._run_stats)
        if self._config.get('verbose'):
            self._logger.info('Time spent: %s seconds' % (time.time() - start_time))
            self._logger.info('instances %(instances)d, solved %(solved)d, unsatisfied clauses %(unsat_clauses)d' % self.last_stats
                              + (' (%(ranks)d ranks, %(backend)s)' % self.last_stats if distributed else ''))

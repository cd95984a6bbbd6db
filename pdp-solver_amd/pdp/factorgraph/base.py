"""Predict driver (reference: src/pdp/factorgraph/base.py, predict path :252-305,451-472).

The train / test loops of the reference's ``FactorGraphTrainerBase`` are out of scope;
``predict`` keeps its signature.  There is no ``nn.DataParallel`` wrap (it mis-scatters ``graph_map``, SURVEY.md
App. B-13): multi-GPU runs shard instances across ranks instead (pdp/parallel.py).
"""

import time

import torch

from pdp import native
from pdp.factorgraph.dataset import FactorGraphDataset


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

    def _predict_epoch(self, validation_loader, post_processor, batch_replication, file):
        with torch.no_grad():
            for data in validation_loader:
                for i in range(len(data[0])):
                    (graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, label, misc_data) = \
                        [self._to_cuda(d[i]) for d in data]
                    self._predict_batch(graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat,
                                        label, misc_data, post_processor, batch_replication, file)

    def _predict_batch(self, graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat, label, misc_data,
                       post_processor, batch_replication, file):
        "reference: base.py:280-305"
        for model in self._model_list:
            state = model.get_init_state(graph_map, batch_variable_map, batch_function_map, edge_feature, graph_feat,
                                         randomized=False, batch_replication=batch_replication)
            prediction, _ = model(init_state=state, graph_map=graph_map, batch_variable_map=batch_variable_map,
                                  batch_function_map=batch_function_map, edge_feature=edge_feature, meta_data=graph_feat,
                                  is_training=False, iteration_num=self._config['test_recurrence_num'],
                                  check_termination=self._check_recurrence_termination, batch_replication=batch_replication)
            if post_processor is not None and callable(post_processor):
                message = post_processor(model, prediction, graph_map, batch_variable_map, batch_function_map,
                                         edge_feature, graph_feat, label, misc_data)
                print(message, file=file)

    def _check_recurrence_termination(self, active, prediction, sat_problem):
        pass

    def predict(self, test_list, out_file, import_path_base=None, post_processor=None, batch_replication=1):
        "Produces predictions for a (trained) PDP model (reference: base.py:451-472)."
        test_loader = FactorGraphDataset.get_loader(
            input_file=test_list, limit=self._config['test_batch_limit'], hidden_dim=self._config['hidden_dim'],
            batch_size=self._config['batch_size'], shuffle=False, num_workers=0,
            max_cache_size=self._config.get('max_cache_size', 100000), batch_replication=batch_replication)
        if import_path_base is not None:
            self._load(import_path_base)
        start_time = time.time()
        self._predict_epoch(test_loader, post_processor, batch_replication, out_file)
        torch.cuda.synchronize()
        if self._config.get('verbose'):
            self._logger.info('Time spent: %s seconds' % (time.time() - start_time))

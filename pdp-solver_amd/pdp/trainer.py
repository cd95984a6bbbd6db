"""Model factory + prediction post-processing for PDP SAT solvers (reference: src/pdp/trainer.py).

The inference side of ``SatFactorGraphTrainer``: ``_build_graph`` (model_type -> solver class,
trainer.py:48-99), ``_check_recurrence_termination`` (:150-162), ``_post_process_predictions`` (:125-148) and the test-mode
metrics ``_compute_evaluation_metrics`` (:108-123: accuracy / recall errors of the clause check and the energy loss), and the training
loss ``_compute_loss`` (:100-106) used by ``FactorGraphTrainerBase._train_batch``.
"""

import numpy as np
import torch
import torch.nn as nn

from pdp.factorgraph.base import FactorGraphTrainerBase
from pdp.nn import solver, util


class Perceptron(nn.Module):
    "1-hidden-layer perceptron with sigmoid output (reference: trainer.py:20-29); classifier head of the neural predictor."

    def __init__(self, input_dimension, hidden_dimension, output_dimension):
        super(Perceptron, self).__init__()
        self._layer1 = nn.Linear(input_dimension, hidden_dimension)
        self._layer2 = nn.Linear(hidden_dimension, output_dimension, bias=False)

    def forward(self, inp):
        return torch.sigmoid(self._layer2(torch.relu(self._layer1(inp))))


class SatFactorGraphTrainer(FactorGraphTrainerBase):
    "Builds a PDP SAT solver from a config dict and runs prediction (reference: trainer.py:34-162)."

    def __init__(self, config, use_cuda, logger):
        super(SatFactorGraphTrainer, self).__init__(config=config, has_meta_data=False, error_dim=config.get('error_dim', 3),
                                                    loss=None, evaluator=nn.L1Loss(), use_cuda=use_cuda, logger=logger)
        self._eps = 1e-8 * torch.ones(1, device=self._device)
        self._loss_evaluator = util.SatLossEvaluator(alpha=self._config.get('exploration', 0.0), device=self._device)
        self._cnf_evaluator = util.SatCNFEvaluator(device=self._device)
        self._counter = 0
        self._max_coeff = 10.0

    def _build_graph(self, config):
        rng = config.get('rng', 'torch')
        seed = int(config.get('random_seed', 0) or 0)
        t = config['model_type']
        common = dict(local_search_iterations=config['local_search_iteration'], epsilon=config['epsilon'], rng=rng, seed=seed)
        if t == 'p-d-p':
            model = solver.SurveyPropagatorSolver(device=self._device, name=config['model_name'], tolerance=config['tolerance'],
                                                  t_max=config['t_max'], persistent=config.get('persistent', True), **common)
            model._isolated = bool(config.get('isolated', False))
        elif t == 'walk-sat':
            model = solver.WalkSATSolver(device=self._device, name=config['model_name'],
                                         iteration_num=config['local_search_iteration'], epsilon=config['epsilon'], rng=rng, seed=seed)
        elif t == 'reinforce':
            model = solver.ReinforceSurveyPropagatorSolver(device=self._device, name=config['model_name'], pi=config['pi'],
                                                           decimation_probability=config['decimation_probability'],
                                                           persistent=config.get('persistent', True), **common)
        elif t in ('np-nd-np', 'np-d-np', 'p-nd-np'):
            model = solver.build_neural_solver(self._device, config, Perceptron, common)
            if hasattr(model._propagator, '_drop_out'):
                # where the training path's dropout masks come from: the device generator (default), or -- dropout_rng: 'torch' -- the global
                # CPU stream drawn exactly as the reference's --cpu_mode run draws it (the golden tests; it builds every mask on the host)
                model._propagator._rng = config.get('dropout_rng', 'device')
            # random initial states (training / test mode): the reference's CPU stream by default, 'device' for throughput
            for plug_in in (model._propagator, model._decimator):
                plug_in._init_rng = config.get('init_rng', 'torch')
            if config.get('dropout', 0) or config.get('init_rng', 'torch') != 'torch':
                # said once: a YAML with `rng: torch` and a seed no longer pins these two streams to the reference's --cpu_mode draws by itself
                self._logger.info("random sources: dropout masks from %r (config key dropout_rng), random initial states from %r (init_rng), random fill / "
                                  "Walk-SAT from %r (rng); 'torch' = the reference's global CPU stream" % (model._propagator._rng if hasattr(model._propagator, '_rng') else 'n/a',
                                                                                                        config.get('init_rng', 'torch'), rng))
        else:
            raise KeyError("unknown model_type %r" % (t,))
        if config.get('verbose'):
            self._logger.info("The model parameter count is %d." % model.parameter_count())
        return [model]

    def _compute_loss(self, model, loss, prediction, label, graph_map, batch_variable_map, batch_function_map, edge_feature, meta_data):
        "the energy of the prediction (reference: trainer.py:100-106); differentiable with respect to the prediction"
        return self._loss_evaluator(variable_prediction=prediction[0], label=label, graph_map=graph_map, batch_variable_map=batch_variable_map,
                                    batch_function_map=batch_function_map, edge_feature=edge_feature, meta_data=meta_data,
                                    global_step=model._global_step, eps=self._eps, max_coeff=self._max_coeff,
                                    loss_sharpness=self._config['loss_sharpness'], sat_problem=getattr(model, '_last_problem', None))

    def _compute_evaluation_metrics(self, model, evaluator, prediction, label, graph_map, batch_variable_map, batch_function_map,
                                    edge_feature, meta_data):
        """[accuracy error, recall error, energy loss] of a prediction on a labelled batch (reference: trainer.py:108-123).
        The clause check and the loss run on the problem the model has just solved (no mask rebuild)."""
        sat_problem = getattr(model, '_last_problem', None)
        output, _ = self._cnf_evaluator(variable_prediction=prediction[0], graph_map=graph_map, batch_variable_map=batch_variable_map,
                                        batch_function_map=batch_function_map, edge_feature=edge_feature, meta_data=meta_data,
                                        sat_problem=sat_problem)
        output = (output.reshape(label.shape) > 0.5).float()
        recall = torch.sum(label * (output - label).abs()) / torch.max(torch.sum(label), self._eps)
        accuracy = evaluator(output, label).unsqueeze(0)
        loss_value = self._loss_evaluator(variable_prediction=prediction[0], label=label, graph_map=graph_map,
                                          batch_variable_map=batch_variable_map, batch_function_map=batch_function_map,
                                          edge_feature=edge_feature, meta_data=meta_data, global_step=model._global_step,
                                          eps=self._eps, max_coeff=self._max_coeff, loss_sharpness=self._config['loss_sharpness'],
                                          sat_problem=sat_problem).unsqueeze(0)
        return torch.cat([accuracy, recall.reshape(1), loss_value], 0)

    def _post_process_predictions(self, model, prediction, graph_map, batch_variable_map, batch_function_map,
                                  edge_feature, graph_feat, label, misc_data):
        """JSON result rows (reference: trainer.py:125-148).  The reference scans ``batch_variable_map == i`` for every
        instance (O(B*V)); instance slices come from one prefix sum here."""
        sat_problem = getattr(model, '_last_problem', None)
        solved, unsat = self._cnf_evaluator(prediction[0], graph_map, batch_variable_map, batch_function_map, edge_feature,
                                            graph_feat, sat_problem=sat_problem)
        output = solved.detach().cpu().numpy()
        unsat_clause_num = unsat.detach().cpu().numpy()
        labs = label.detach().cpu().numpy()
        bits = (prediction[0].detach().reshape(-1) > 0.5).to(torch.uint8).cpu().numpy()
        counts = np.bincount(batch_variable_map.detach().cpu().numpy().astype(np.int64), minlength=output.shape[0])
        offs = np.concatenate(([0], np.cumsum(counts)))
        rows = []
        for i in range(output.shape[0]):
            instance = {
                'ID': misc_data[i][0] if len(misc_data[i]) > 0 else "",
                'label': int(labs[i, 0]),
                'solved': int(output[i].flatten()[0] == 1),
                'unsat_clauses': int(unsat_clause_num[i].flatten()[0]),
                'solution': bits[offs[i]:offs[i + 1]].astype(int).tolist()
            }
            rows.append(str(instance).replace("'", '"') + "\n")
            self._counter += 1
        if hasattr(self, '_run_stats'):
            self._run_stats[0] += int(output.shape[0]); self._run_stats[1] += int((output.reshape(output.shape[0], -1)[:, 0] == 1).sum())
            self._run_stats[2] += int(unsat_clause_num.sum())
        return "".join(rows)

    def _check_recurrence_termination(self, active, prediction, sat_problem):
        "De-activates the instances the model has already solved (reference: trainer.py:150-162)."
        sat_problem._native.check_termination(active.reshape(-1), prediction[0].reshape(-1).contiguous())

    _check_recurrence_termination._pdp_standard_termination = True

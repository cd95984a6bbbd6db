#!/usr/bin/env python3
"""The PDP trainer / tester (reference: src/satyr-train-test.py).

``python satyr-train-test.py <config.yaml> [-t] [-l best|last] [-r] [-g] [-b R]``: without ``-t`` the configured neural solver
(model_type np-nd-np) is trained -- Adam on the energy loss, from the JSON training set or, with ``-g``, from the configured CNF
generator -- with a validation pass per epoch and best / last checkpoints under ``model_path``; then (and with ``-t`` only) the solver
is evaluated on the labelled JSON datasets of ``test_path``: accuracy / recall per dataset plus the two result CSV files next to a
dataset directory, like the reference.  ``-c`` (CPU mode) is rejected: the hot path runs on the MI355X only.
"""

import argparse
import csv
import logging
import os
import sys

import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from pdp.trainer import SatFactorGraphTrainer


def write_results(result_list, file_path, column):
    "one row per dataset: [file, value]; column 1 = recall error of model 0 (reference: write_to_csv), 2 = seconds (write_to_csv_time)"
    with open(file_path, mode='w', newline='') as f:
        writer = csv.writer(f, delimiter=',', quotechar='"', quoting=csv.QUOTE_MINIMAL)
        for row in result_list:
            writer.writerow([row[0], row[1][1, 0] if column == 1 else row[2]])


def make_generator(config):
    "the CNF generator a training run draws its examples from (reference: satyr-train-test.py:85-95)"
    from pdp import cnf_generators as G
    if config['generator'] == 'modular':
        return G.ModularCNFGenerator(config['min_k'], config['min_n'], config['max_n'], config['min_q'], config['max_q'], config['min_c'],
                                     config['max_c'], config['min_alpha'], config['max_alpha'])
    if config['generator'] == 'v-modular':
        return G.VariableModularCNFGenerator(config['min_k'], config['max_k'], config['min_n'], config['max_n'], config['min_q'], config['max_q'],
                                             config['min_c'], config['max_c'], config['min_alpha'], config['max_alpha'])
    return G.UniformCNFGenerator(config['min_n'], config['max_n'], config['min_k'], config['max_k'], config['min_alpha'], config['max_alpha'])


def run(random_seed, config_file, is_training, load_model, cpu, reset_step, use_generator, batch_replication):
    if not use_generator:
        np.random.seed(random_seed)
        torch.manual_seed(random_seed)
    with open(config_file, 'r') as f:
        config = yaml.safe_load(f)
    logging.basicConfig(level=logging.DEBUG, format='[%(levelname)s] %(asctime)s - %(name)s: %(message)s')
    logger = logging.getLogger(config['model_name'] + ' (' + str(config['version']) + ')')
    for key in ('train_path', 'validation_path'):
        if is_training and not isinstance(config.get(key), list):
            config[key] = [os.path.join(config[key], f) for f in os.listdir(config[key])
                           if os.path.isfile(os.path.join(config[key], f)) and f.endswith('.json')]
    base = os.path.join(os.path.relpath(config['model_path']), config['model_name'], str(config['version']))
    best_base, last_base = os.path.join(base, 'best'), os.path.join(base, 'last')
    if is_training:
        # a training run draws its random initial states on the device unless the YAML says otherwise (init_rng: 'torch' = the reference's
        # CPU stream, which costs seconds per batch at a million edges; dropout masks come from the device generator by default as well)
        config.setdefault('init_rng', 'device')
    trainer = SatFactorGraphTrainer(config=config, use_cuda=not cpu, logger=logger)
    if is_training:
        why = trainer.UNTRAINABLE.get(config['model_type'])
        if why is not None:
            raise SystemExit("satyr-train-test.py: model_type %r cannot be trained: %s" % (config['model_type'], why))
        import torch.optim as optim
        for d in (best_base, last_base):
            os.makedirs(d, exist_ok=True)
        if config['verbose']:
            logger.info("Starting the training phase...")
        trainer.train(train_list=config['train_path'], validation_list=config['validation_path'],
                      optimizer=optim.Adam(trainer.get_parameter_list(), lr=config['learning_rate'], weight_decay=config['weight_decay']),
                      last_export_path_base=last_base, best_export_path_base=best_base, metric_index=config['metric_index'],
                      load_model=load_model, reset_step=reset_step, generator=make_generator(config) if use_generator else None,
                      train_epoch_size=config.get('train_epoch_size', 0))
    import_path_base = os.path.join(base, load_model) if load_model in ('last', 'best') else None
    if config['verbose']:
        logger.info("Starting the test phase...")
    results = []
    for test_files in config['test_path']:
        if config['verbose']:
            logger.info("Testing " + test_files)
        result = trainer.test(test_list=test_files, import_path_base=import_path_base, batch_replication=batch_replication)
        if config['verbose']:
            for filename, errors, _ in result:
                print('Dataset: ' + filename)
                print("Accuracy: \t%s" % (1 - errors[0]))
                print("Recall: \t%s" % (1 - errors[1]))
        if os.path.isdir(test_files):
            stem = os.path.join(test_files, config['model_type'] + '_' + config['model_name'] + '_' + str(config['version']))
            write_results(result, stem + '-results.csv', 1)
            write_results(result, stem + '-results-time.csv', 2)
        results += result
    return results


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('config', help='The configuration YAML file')
    parser.add_argument('-t', '--test', help='The test mode', action='store_true')
    parser.add_argument('-l', '--load_model', help='Load the previous model')
    parser.add_argument('-c', '--cpu_mode', help='Run on CPU (not available)', action='store_true')
    parser.add_argument('-r', '--reset', help='Reset the global step', action='store_true')
    parser.add_argument('-g', '--use_generator', help='Use a generator (training only)', action='store_true')
    parser.add_argument('-b', '--batch_replication', help='Batch replication factor', type=int, default=1)
    args = parser.parse_args()
    run(0, args.config, not args.test, args.load_model, args.cpu_mode, args.reset, args.use_generator, args.batch_replication)

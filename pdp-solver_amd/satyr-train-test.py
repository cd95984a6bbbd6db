#!/usr/bin/env python3
"""Test mode of the PDP trainer/tester (reference: src/satyr-train-test.py with ``-t``).

``python satyr-train-test.py <config.yaml> -t [-l best|last] [-b R]`` evaluates the configured solver on the labelled JSON datasets of
``test_path`` and prints accuracy / recall per dataset (plus the two result CSV files next to a dataset directory), like the
reference.  Training (the mode without ``-t``) is out of scope of this build and is rejected; ``-c`` (CPU mode) too -- the hot path
runs on the MI355X only.
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


def run(random_seed, config_file, is_training, load_model, cpu, reset_step, use_generator, batch_replication):
    if is_training:
        raise SystemExit("satyr-train-test.py: training is out of scope of this build (SURVEY.md section 8 f3); use -t for the test mode")
    np.random.seed(random_seed)
    torch.manual_seed(random_seed)
    with open(config_file, 'r') as f:
        config = yaml.safe_load(f)
    logging.basicConfig(level=logging.DEBUG, format='[%(levelname)s] %(asctime)s - %(name)s: %(message)s')
    logger = logging.getLogger(config['model_name'] + ' (' + str(config['version']) + ')')
    trainer = SatFactorGraphTrainer(config=config, use_cuda=not cpu, logger=logger)
    base = os.path.join(os.path.relpath(config['model_path']), config['model_name'], str(config['version']))
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

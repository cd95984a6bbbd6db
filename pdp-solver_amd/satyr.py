#!/usr/bin/env python3
"""Run a PDP solver against a test set on the MI355X (drop-in for the reference's src/satyr.py).

Same positionals, flags, YAML keys and output format as the reference CLI (reference: satyr.py:45-109).
Additions: ``--rng {torch,philox}`` (torch = the reference's CPU random stream, bit-compatible results for the
same ``-s`` seed; philox = on-device counters, fastest), ``--stepwise`` (disable the one-launch persistent loop) and ``--isolated``
(every instance on its own instead of the reference's batch-wide couplings).
``-c/--cpu_mode`` is rejected: the hot path has no CPU fallback.  Launched through ``python -m torch.distributed.run --nproc-per-node N``
it runs one process per GPU on a shard of the input each and reduces the result once over RCCL.
"""

import argparse
import logging
import os
import sys
from datetime import datetime

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import dimacs2json  # noqa: E402
from pdp.trainer import SatFactorGraphTrainer  # noqa: E402


def _open_output(path):
    "where the result rows go: stdout, a file, or nowhere (ranks other than 0 of a sharded run: rank 0 writes the gathered rows)"
    if int(os.environ.get('RANK', '0')) != 0:
        return open(os.devnull, 'w'), True
    if not path:
        return sys.stdout, False
    return open(path, 'w'), True


def run(config, logger, output):
    "Seeds the two random sources, builds the model of config['model_type'] and writes one result row per input instance."
    seed = config['random_seed']
    np.random.seed(seed)
    torch.manual_seed(seed)
    say = logger.info if config['verbose'] else (lambda *a, **k: None)
    say("Building the computational graph...")
    solver = SatFactorGraphTrainer(config=config, use_cuda=not config['cpu_mode'], logger=logger)
    solver._counter = 0
    say("Starting the prediction phase...")
    sink, owned = _open_output(output)
    try:
        solver.predict(test_list=config['test_path'], out_file=sink, import_path_base=config['model_path'],
                       post_processor=solver._post_process_predictions, batch_replication=config['batch_replication'])
    finally:
        if owned:
            sink.close()
    return solver


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('model_config', help='The model configuration yaml file')
    parser.add_argument('test_path', help='The input test path')
    parser.add_argument('test_recurrence_num', help='The number of iterations for the PDP', type=int)
    parser.add_argument('-b', '--batch_replication', help='Batch replication factor', type=int, default=1)
    parser.add_argument('-z', '--batch_size', help='Batch size', type=int, default=5000)
    parser.add_argument('-m', '--max_cache_size', help='Maximum cache size', type=int, default=100000)
    parser.add_argument('-l', '--test_batch_limit', help='Memory limit for mini-batches', type=int, default=40000000)
    parser.add_argument('-w', '--local_search_iteration', help='Number of iterations for post-processing local search', type=int, default=100)
    parser.add_argument('-e', '--epsilon', help='Epsilon probablity for post-processing local search', type=float, default=0.5)
    parser.add_argument('-v', '--verbose', help='Verbose', action='store_true')
    parser.add_argument('-c', '--cpu_mode', help='Run on CPU (not available in the MI355X build)', action='store_true')
    parser.add_argument('-d', '--dimacs', help='The input folder contains DIMACS files', action='store_true')
    parser.add_argument('-s', '--random_seed', help='Random seed', type=int, default=int(datetime.now().microsecond))
    parser.add_argument('-o', '--output', help='The JSON output file', default='')
    parser.add_argument('--rng', help='Random source for random fill / Walk-SAT', choices=['torch', 'philox'], default='torch')
    parser.add_argument('--stepwise', help='Disable the persistent one-launch PDP loop', action='store_true')
    parser.add_argument('--isolated', help='Solve every instance on its own: none of the batch-wide couplings of the reference '
                        '(global minima, NaN poisoning of the whole batch); p-d-p only, results differ from the reference where those couplings act.  On several ranks (torch.distributed.run) '
                        'the instances of every forward are then spread over all GPUs',
                        action='store_true')
    parser.add_argument('--split-forward', dest='split_forward', help='On several ranks: spread EVERY forward over all GPUs (one contiguous instance range '
                        'per rank) and keep the couplings of the reference -- its batch-wide reductions are completed across the ranks chunk by chunk; '
                        'p-d-p, -b 1; the rows are those of the single-process run', action='store_true')
    args = vars(parser.parse_args(argv))

    with open(args['model_config'], 'r') as f:
        model_config = yaml.safe_load(f)

    fmt = '[%(levelname)s] %(asctime)s - %(name)s: %(message)s'
    logging.basicConfig(level=logging.DEBUG, format=fmt)
    logger = logging.getLogger(model_config['model_name'])

    # -d: the reference converts the DIMACS input into a temporary JSON file first (satyr.py:66-80); here the loader reads the
    # DIMACS files directly through the native parser (same instances, same order, same labels -- dataset.dimacs_file_list)
    temp_file_name = None
    if args['dimacs'] and args['verbose']:
        logger.info("Reading DIMACS files...")

    config = {**model_config, **args}
    if config['model_type'] in ('p-d-p', 'walk-sat', 'reinforce'):
        config['model_path'] = None
        config['hidden_dim'] = 3
    if config['model_type'] == 'walk-sat':
        config['local_search_iteration'] = config['test_recurrence_num']
    config['dropout'] = 0
    config['error_dim'] = 1
    config['exploration'] = 0
    config['persistent'] = not config['stepwise']

    # one process per GPU under torch.distributed.run: instances are sharded across the ranks (pdp/factorgraph/base.py::predict)
    # PDP_DIST_FORCE=1 joins the process group at world size 1 as well (torch.distributed.run --nproc-per-node 1): the all-reduce and the
    # row gather then run through RCCL on a one-GPU box exactly as they do on eight
    world = int(os.environ.get('WORLD_SIZE', '1'))
    grouped = world > 1 or (os.environ.get('PDP_DIST_FORCE') == '1' and 'RANK' in os.environ)
    if grouped:
        import torch.distributed as dist
        # PDP_DIST_BACKEND=gloo lets several ranks share one GPU (checks of the sharded path on a single-GPU box); RCCL needs one GPU each
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=os.environ.get('PDP_DIST_BACKEND', 'nccl'))
    try:
        run(config, logger, config['output'])
    finally:
        if temp_file_name is not None and os.path.exists(temp_file_name):
            os.remove(temp_file_name)
        if grouped:
            import torch.distributed as dist
            dist.destroy_process_group()
    if world == 1 or int(os.environ.get('RANK', '0')) == 0:
        print('')


if __name__ == '__main__':
    main()

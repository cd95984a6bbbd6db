"""Batch-sharded data parallelism: one process per GPU, no collective on the data path.

The reference has no working multi-GPU path (``nn.DataParallel`` would scatter ``graph_map`` along dim 0, SURVEY.md
App. B-13).  No message crosses an instance boundary, but the reference couples the instances of one *loader batch* (base.py:252-278
loops the DataLoader's batches, dataset.py:189-211): batch-global minima inside sparse_max / argmax, one NaN survey that stops the
decimation of the whole batch, the dynamic segments cut from the batch's edge counts (dataset.py:36-72).  The unit
that is dealt to ranks is therefore the WHOLE LOADER BATCH: the loader forms the same batches (and the same segments) whatever the
rank count, rank r solves a contiguous range of them (``deal_batches``, balanced by input size) completely on its own GPU, random
numbers are keyed by the global batch / segment index (``batch_seed``), and the ranks meet exactly once, in an all-reduce(sum) of
``[instances, solved, unsat clauses]`` -- in test mode of the metric sums ``[accuracy, recall, loss]`` and the example count -- (RCCL
over xGMI on a node: ``backend='nccl'``; the tests use ``gloo``).  Result rows are gathered in rank order, which is batch order.  An
N-rank run therefore writes exactly the rows of the 1-rank run (tests/test_parallel_gloo.py, tests/test_sharded_gpu.py).
BASELINE configs[3]: 40 000 instances with ``-z 5000`` are 8 loader batches, one per GPU.

``shard_bounds`` / ``shard_items`` cut ONE batch by instances; only bench.py uses that (its synthetic batch has no loader and is timed
in ``--isolated``-equivalent weak scaling: every rank generates its own B instances).
"""

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(edge_counts, world_size):
    """Contiguous instance ranges [lo, hi) per rank, balanced by the cumulative edge count.

    Every rank gets at least one instance while instances remain; ranges are contiguous so replicas / result
    order stay rank-major."""
    edge_counts = np.asarray(edge_counts, dtype=np.int64)
    n = int(edge_counts.size)
    world_size = int(world_size)
    if n == 0:
        return [(0, 0)] * world_size
    csum = np.cumsum(edge_counts)
    total = int(csum[-1])
    bounds, lo = [], 0
    for r in range(world_size):
        later_ranks = world_size - 1 - r
        if later_ranks == 0:
            hi = n
        else:
            # smallest prefix whose edge count reaches this rank's share of the total ...
            hi = int(np.searchsorted(csum, total * (r + 1) / float(world_size), side='left')) + 1
            # ... but at least one instance while instances remain, and one left over for each later rank when there are enough
            hi = max(hi, lo + 1)
            hi = min(hi, max(lo + 1, n - later_ranks), n)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def deal_batches(batch_weights, world_size):
    """Contiguous ranges [lo, hi) of loader-batch indices per rank, balanced by the batches' weights (input bytes as the proxy of the
    edge count -- no rank parses another rank's instances).  Contiguous, so the rank-ordered gather of the rows is the loader's order.
    With fewer batches than ranks the last ranks get an empty range: a batch is the reference's coupling domain and is never split."""
    return shard_bounds(batch_weights, world_size)


def batch_seed(seed, batch_index, segment_index=0):
    """The 64-bit Philox key of segment ``segment_index`` of loader batch ``batch_index`` (global indices, the same on every rank
    count).  (0, 0) keeps the run's seed, so a single-batch run draws what a direct ``forward`` call with that seed draws."""
    return (int(seed) + 0x9E3779B97F4A7C15 * int(batch_index) + 0xC2B2AE3D27D4EB4F * int(segment_index)) & 0xFFFFFFFFFFFFFFFF


def shard_items(items, rank, world_size):
    "the loader items (dataset.parse_line tuples) this rank owns"
    lo, hi = shard_bounds([it[2].shape[1] for it in items], world_size)[rank]
    return items[lo:hi], lo


def reduce_stats(n_instances, n_solved, n_unsat_clauses, device=None, group=None):
    """The single collective of the path: all-reduce(sum) of three counters.  Works without an initialised process
    group (world size 1)."""
    t = torch.tensor([float(n_instances), float(n_solved), float(n_unsat_clauses)], dtype=torch.float64,
                     device=device if device is not None else 'cpu')
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    n, s, u = [float(x) for x in t.tolist()]
    return dict(instances=int(n), solved=int(s), unsat_clauses=int(u), solved_fraction=(s / n if n else 0.0))


def reduce_test_metrics(error_sums, n_examples, device=None, group=None):
    """Test mode across ranks: every rank holds the per-example-weighted sums of its shard ([accuracy error, recall error, loss] x
    models, ``FactorGraphTrainerBase._last_test_counts``); one all-reduce(sum) of the sums and the example count gives the means the
    single-process run reports (base.py:219)."""
    sums = np.asarray(error_sums, dtype=np.float64)
    t = torch.tensor(list(sums.reshape(-1)) + [float(n_examples)], dtype=torch.float64, device=device if device is not None else 'cpu')
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    vals = t.cpu().numpy()
    n = float(vals[-1])
    return (vals[:-1].reshape(sums.shape) / n if n else vals[:-1].reshape(sums.shape)), int(n)


def gather_rows(rows, group=None):
    "rank-ordered list of every rank's result rows (python objects; host side only)"
    if not (dist.is_available() and dist.is_initialized()):
        return list(rows)
    world = dist.get_world_size(group)
    out = [None] * world
    dist.all_gather_object(out, list(rows), group=group)
    return [r for part in out for r in part]


def solve_sharded(batches, solve_fn, rank=None, world_size=None, device=None):
    """``batches``: the loader batches of the run (lists of loader items), the same list on every rank.  Runs
    ``solve_fn(items, batch_index) -> (solved [b], unsat [b], rows list)`` on the batches dealt to this rank and reduces the counters;
    returns (stats, all rows in loader order, this rank's [lo, hi) batch range).  ``solve_fn`` is the native forward in production and
    the CPU oracle in the gloo tests."""
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = deal_batches([sum(it[2].shape[1] for it in b) for b in batches], world_size)[rank]
    n = n_solved = n_unsat = 0
    rows = []
    for j in range(lo, hi):
        solved, unsat, r = solve_fn(batches[j], j)
        n += len(batches[j]); n_solved += float(np.sum(solved)); n_unsat += float(np.sum(unsat)); rows += list(r)
    stats = reduce_stats(n, n_solved, n_unsat, device=device)
    return stats, gather_rows(rows), (lo, hi)

"""Segment-sharded data parallelism: one process per GPU, no collective on the data path.

The reference has no working multi-GPU path (``nn.DataParallel`` would scatter ``graph_map`` along dim 0, SURVEY.md
App. B-13).  No message crosses an instance boundary, but the reference couples the instances of one ``forward`` call: batch-global
minima inside sparse_max / argmax, one NaN survey that stops the decimation of everything in the call.  One
``forward`` is one dynamic SEGMENT of a loader batch (base.py:252-278 loops ``for i in range(len(data[0]))`` over the segments
``DynamicBatchDivider.divide`` cut from the batch's edge counts, dataset.py:24-74) -- so the segment, not the loader batch, is the unit
that may move between GPUs without changing a row.  Every rank enumerates the same (loader batch, segment) units (the loader forms the
batches and cuts the segments from the edge counts alone, without collating anything), the units of a batch are dealt by
longest-processing-time-first on their edge counts onto the least loaded rank (``deal_units``; the loads carry over from batch to
batch, so a run of many one-segment batches and a run of one batch with many segments both keep every GPU busy), random numbers are
keyed by the global (batch, segment) index (``batch_seed``), and the ranks meet exactly once, in an all-reduce(sum) of ``[instances,
solved, unsat clauses]`` -- in test mode of the metric sums ``[accuracy, recall, loss]`` and the example count -- (RCCL over xGMI on a
node: ``backend='nccl'``; the CPU tests use ``gloo``).  Result rows travel with their unit index and rank 0 writes them in unit order,
which is the single-process order.  An N-rank run therefore writes exactly the rows of the 1-rank run (tests/test_parallel_gloo.py,
tests/test_sharded_gpu.py).  BASELINE configs[3]: 40 000 instances with ``-z 5000`` are 8 one-segment batches, one per GPU;
configs[4] (dynamic batching, ``-b 4``): a loader batch falls into many segments, which spread over the 8 GPUs.

``--isolated`` (every instance solved on its own: none of the couplings above) makes the INSTANCE the unit: every segment is then cut
into one contiguous instance range per rank, balanced by edges (``shard_bounds``) -- a *part* -- so a run of fewer segments than GPUs,
down to one forward, uses every GPU.  The Philox counters of a part start at the part's first variable / instance inside its segment
(``pdp_problem_set_rng_base``), so an instance draws what it draws when the segment is solved whole and the N-rank rows are the 1-rank
``--isolated --rng philox`` rows (tests/test_sharded_gpu.py).  Batch replication keeps segment dealing (replica r of variable v has index
v + r V: no contiguous base).  One batch-global quantity is left in Walk-SAT -- the minimum of a step's candidate vector inside
util.sparse_argmax, 0 whenever any variable of the forward is not in an unsatisfied clause -- a part whose every variable is in an
unsatisfied clause while the whole segment has one that is not would round one comparison differently; not observed, not excluded.
The strict (coupled) semantics across GPUs: ``--split-forward`` cuts the segments the same way and completes the reference's batch-wide
reductions across the parts -- per chunk of sweeps of the persistent solver one small all-gather of its control words (first NaN sweep: min,
exact-zero record of the batch-global minimum: or, executed sweeps: max), one more after a poison replay, one for the Walk-SAT record
(``make_exchange`` -> ``native.Problem.set_exchange`` -> C ABI ``pdp_problem_set_exchange``); the rows are those of the 1-rank strict run.
A segment whose speculation fails (small batches: no variable supplies the exact zero of the batch-global minimum) is solved whole by the
rank of its first part with the single-process loops -- the parts agree on the outcome before any later collective.

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


def deal_units(weights, world_size, loads=None):
    """Owner rank of each unit (segment) of one loader batch: the units are taken heaviest first (ties: lowest index) and each goes to the
    rank with the smallest load so far (ties: lowest rank).  ``loads`` (a list of ``world_size`` numbers, updated in place) carries the
    load from batch to batch.  Pure function of its arguments: every rank computes the same table, nothing is exchanged."""
    world_size = int(world_size)
    if loads is None:
        loads = [0] * world_size
    owners = [0] * len(weights)
    for u in sorted(range(len(weights)), key=lambda k: (-int(weights[k]), k)):
        r = min(range(world_size), key=lambda k: (loads[k], k))
        owners[u] = r
        loads[r] += int(weights[u])
    return owners


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


def make_exchange(device=None, group=None):
    """The callback of a coupled forward spread over the ranks (native.Problem.set_exchange): element-wise min / max / bit-wise OR of three
    small uint32 arrays over all ranks, in ONE collective (all_gather_into_tensor of the concatenated words).  Under nccl (RCCL) the words
    go host -> device -> all ranks -> host through two pinned staging tensors that live as long as the callback: two small asynchronous copies
    and ONE stream synchronisation per exchange (the library needs the merged words on the host: its launches depend on them); under gloo
    the collective runs on the host.  ``exchange.calls`` / ``exchange.seconds`` count what the run spent here."""
    staged = {}

    def exchange(mins, maxs, ors):
        import time
        t0 = time.perf_counter()
        n = (mins.size, maxs.size, ors.size)
        total = n[0] + n[1] + n[2]
        world = dist.get_world_size(group)
        on_gpu = dist.get_backend(group) == 'nccl'
        key = (total, world, on_gpu)
        if key not in staged:
            staged.clear()
            if on_gpu:
                staged[key] = (torch.empty(total, dtype=torch.int32).pin_memory(), torch.empty(world * total, dtype=torch.int32).pin_memory(),
                               torch.empty(total, dtype=torch.int32, device=device), torch.empty(world * total, dtype=torch.int32, device=device))
            else:
                staged[key] = (torch.empty(total, dtype=torch.int32), torch.empty(world * total, dtype=torch.int32), None, None)
        h_in, h_out, d_in, d_out = staged[key]
        h_in.numpy()[:] = np.concatenate((mins, maxs, ors)).astype(np.uint32).view(np.int32)        # (bit patterns: the reductions below are on uint32)
        if on_gpu:
            d_in.copy_(h_in, non_blocking=True)
            dist.all_gather_into_tensor(d_out, d_in, group=group)
            h_out.copy_(d_out, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        else:
            dist.all_gather_into_tensor(h_out, h_in, group=group)
        every = h_out.numpy().view(np.uint32).reshape(world, total)
        mins[:] = every[:, :n[0]].min(axis=0) if n[0] else mins
        maxs[:] = every[:, n[0]:n[0] + n[1]].max(axis=0) if n[1] else maxs
        if n[2]:
            ors[:] = np.bitwise_or.reduce(every[:, n[0] + n[1]:], axis=0)
        exchange.calls += 1
        exchange.seconds += time.perf_counter() - t0
    exchange.calls, exchange.seconds = 0, 0.0
    return exchange


def gather_rows(rows, group=None, dst=0):
    """Rank-ordered list of every rank's result rows ON RANK ``dst`` (the other ranks get their own rows back: only the writer needs the
    rest).  The rows are pickled once per rank and travel as ONE padded uint8 tensor per rank in one ``gather`` to the writer (RCCL on the
    device under nccl) -- ``all_gather_object`` handed every rank every row: 8 x 8 x 6.5 MB for configs[3] (40 000 rows of 400 values),
    where the writer needs 8 x 6.5 MB once.  The lengths travel first, in one small all-gather."""
    if not (dist.is_available() and dist.is_initialized()):
        return list(rows)
    import pickle
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    on_gpu = dist.get_backend(group) == 'nccl'
    dev = torch.device('cuda', torch.cuda.current_device()) if on_gpu else torch.device('cpu')
    blob = np.frombuffer(pickle.dumps(list(rows), protocol=pickle.HIGHEST_PROTOCOL), dtype=np.uint8)
    sizes = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, torch.tensor([blob.size], dtype=torch.int64, device=dev), group=group)
    sizes = [int(x) for x in sizes.cpu().tolist()]
    cap = max(max(sizes), 1)
    mine = torch.zeros(cap, dtype=torch.uint8, device=dev)
    mine[:blob.size] = torch.from_numpy(blob.copy()).to(dev)
    parts = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(mine, parts, dst=dst, group=group)
    if rank != dst:
        return list(rows)
    out = []
    for r in range(world):
        out += pickle.loads(parts[r][:sizes[r]].cpu().numpy().tobytes())
    return out


def gather_units(units, group=None):
    """``units``: this rank's [((batch, segment), payload), ...].  Returns -- on rank 0; the other ranks get their own -- the payloads of
    all ranks in (batch, segment) order: the order the single-process run produces them in.  A rank without units contributes an empty list."""
    return [payload for _, payload in sorted(gather_rows(units, group), key=lambda kv: kv[0])]


def solve_sharded(batches, solve_fn, rank=None, world_size=None, device=None, limit=None, hidden_dim=1, split_instances=False):
    """``batches``: the loader batches of the run (lists of loader items), the same list on every rank; ``limit`` / ``hidden_dim``: the
    dynamic-batching budget that cuts a batch into segments (None: one segment per batch).  Runs ``solve_fn(items, batch_index,
    segment_index) -> (solved [b], unsat [b], rows list)`` on the units dealt to this rank and reduces the counters; returns (stats, rows,
    this rank's [(batch, segment), ...]) -- `rows` is every rank's rows in single-process order ON RANK 0 ONLY (the writer); the other
    ranks get their own rows back (gather_rows sends the rows to the writer, not to everybody).  ``solve_fn`` is the native forward in production and the CPU
    oracle in the gloo tests.  ``split_instances`` (isolated instances): every segment is cut into one instance range per rank and
    ``solve_fn(items, batch_index, segment_index, first_variable, first_instance)`` solves this rank's part; units are (batch, segment, part)."""
    from pdp.factorgraph import dataset
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    n = n_solved = n_unsat = 0
    units, mine, loads = [], [], [0] * world_size
    for j, batch in enumerate(batches):
        edges = [it[2].shape[1] for it in batch]
        segments = [list(range(len(batch)))] if limit is None else dataset.divide(edges, limit, hidden_dim)
        owners = deal_units([sum(edges[k] for k in seg) for seg in segments], world_size, loads) if not split_instances else None
        for i, seg in enumerate(segments):
            if split_instances:
                lo, hi = shard_bounds([edges[k] for k in seg], world_size)[rank]
                if hi <= lo:
                    continue
                items, key = [batch[k] for k in seg[lo:hi]], (j, i, rank)
                solved, unsat, r = solve_fn(items, j, i, sum(int(batch[k][0]) for k in seg[:lo]), lo)
            else:
                if owners[i] != rank:
                    continue
                items, key = [batch[k] for k in seg], (j, i)
                solved, unsat, r = solve_fn(items, j, i)
            n += len(items); n_solved += float(np.sum(solved)); n_unsat += float(np.sum(unsat))
            units.append((key, list(r))); mine.append(key)
    stats = reduce_stats(n, n_solved, n_unsat, device=device)
    return stats, [row for part in gather_units(units) for row in part], mine

"""Input pipeline: compact-JSON CNF lines -> collated, instance-contiguous edge lists.

Mirrors the reference loader's tensor layout exactly (reference: src/pdp/factorgraph/dataset.py:
_convert_line :120-136, dag_collate_fn :138-187, DynamicBatchDivider.divide :24-74) because that layout
IS the input contract of the hot path: ``graph_map int32 [2,E]`` (row 0 global variable id, row 1 global
clause id), ``batch_variable_map int32 [V]``, ``batch_function_map int32 [F]``, ``edge_feature fp32 [E,1]``.
The implementation is new: vectorised numpy concatenation with prefix offsets instead of the
reference's per-instance np.concatenate growth (quadratic in the batch size), and no torch DataLoader
worker processes (host I/O is out of scope).
"""

import json

import numpy as np
import torch


def parse_line(json_str):
    """One JSON line -> (variable_num, function_num, graph_map[2,e], edge_feature[e], label, misc)."""
    data = json.loads(json_str)
    variable_num, function_num = int(data[0][0]), int(data[0][1])
    signed = np.asarray(data[1], dtype=np.int32)
    variable_ind = np.abs(signed) - 1
    function_ind = np.abs(np.asarray(data[2], dtype=np.int32)) - 1
    edge_feature = np.sign(signed).astype(np.float32)
    graph_map = np.stack((variable_ind, function_ind)).astype(np.int32)
    misc = data[4] if len(data) > 4 else []
    return variable_num, function_num, graph_map, edge_feature, float(data[3]), misc


def instance_from_clauses(n, clauses, label=-1.0, name=""):
    """Build the same tuple directly from a clause list (skips the JSON round trip)."""
    from pdp import generator
    var_num, clause_num, sv, ci = generator.compact_instance(n, clauses)
    graph_map = np.stack((np.abs(sv) - 1, ci - 1)).astype(np.int32)
    return var_num, clause_num, graph_map, np.sign(sv).astype(np.float32), float(label), [name] if name else []


def random_ksat_items(batch, n, k=3, m=None, seed=0):
    """``batch`` uniform random k-SAT instances (instance i from RandomState(seed + i)) as loader items."""
    from pdp import generator
    if m is None:
        m = generator.clause_count(n, k)
    items = []
    for i in range(batch):
        variables, signs = generator.uniform_ksat_arrays(n, m, k, np.random.RandomState(seed + i))
        vn, fn, gm, ef = generator.compact_arrays(n, variables, signs)
        items.append((vn, fn, gm, ef, -1.0, ["rand_%d" % (seed + i)]))
    return items


def divide(edge_nums, limit, hidden_dim):
    """Dynamic batching: index lists of the segments of one loader batch.

    Same policy as the reference (dataset.py:36-72): if ``limit // (max_edges * hidden_dim) >= batch`` the
    batch is one segment in input order; otherwise instances are sorted by edge count (descending, stable)
    and cut greedily with ``allowed = limit // (edges_of_first * hidden_dim)``.  An instance larger than the
    limit gets a segment of its own (the reference loops forever there)."""
    batch = len(edge_nums)
    if batch == 0:
        return []
    if (limit // (max(edge_nums) * hidden_dim)) >= batch:
        return [list(range(batch))]
    order = sorted(range(batch), key=lambda k: edge_nums[k], reverse=True)
    segments, i = [], 0
    while i < batch:
        allowed = max(1, limit // (edge_nums[order[i]] * hidden_dim))
        segments.append(order[i:min(i + allowed, batch)])
        i += allowed
    return segments


def collate_segment(items):
    """Collate instances into one batch (dict of numpy arrays + label/misc lists)."""
    vn = np.asarray([it[0] for it in items], dtype=np.int64)
    fn = np.asarray([it[1] for it in items], dtype=np.int64)
    en = np.asarray([it[2].shape[1] for it in items], dtype=np.int64)
    v_off = np.concatenate(([0], np.cumsum(vn)[:-1]))
    f_off = np.concatenate(([0], np.cumsum(fn)[:-1]))
    gm = np.concatenate([it[2] for it in items], axis=1).astype(np.int64) if len(items) else np.zeros((2, 0), np.int64)
    gm[0] += np.repeat(v_off, en)
    gm[1] += np.repeat(f_off, en)
    return dict(
        graph_map=gm.astype(np.int32),
        batch_variable_map=np.repeat(np.arange(len(items), dtype=np.int32), vn),
        batch_function_map=np.repeat(np.arange(len(items), dtype=np.int32), fn),
        edge_feature=np.concatenate([it[3] for it in items]).astype(np.float32).reshape(-1, 1),
        label=np.asarray([[it[4]] for it in items], dtype=np.float32),
        misc_data=[it[5] for it in items],
        batch_size=len(items))


class SegmentList(list):
    "the collated segments of one loader batch this process solves + where they sit in the run"
    batch_index = 0           # global index of the loader batch
    segment_ids = ()          # global segment index (inside the batch) of every entry
    parts = None              # isolated instances dealt to ranks: per entry (part index, first variable, first instance) inside its segment
    whole = None              # --split-forward: per entry a callable that collates the WHOLE segment (the single-process fallback of a failed speculation)


class LoaderBatch(tuple):
    """What the loader yields: the reference's 7-tuple of per-segment lists (dataset.py:138-187) plus ``index`` (global loader-batch index)
    and ``segments`` (global segment index of every entry) -- the key of the device-side random numbers and of the row order."""
    index = 0
    segments = ()
    parts = None              # see SegmentList.parts
    whole = None              # see SegmentList.whole


def json_edge_count(line):
    "number of edges of a compact-JSON instance line without parsing it: the length of its second list"
    try:
        s = line.index('], [') + 4
        e = line.index(']', s)
    except ValueError:
        return parse_line(line)[2].shape[1]
    return line.count(',', s, e) + 1 if line[s:e].strip() else 0


def json_variable_count(line):
    "number of variables of a compact-JSON instance line without parsing it: the first number of its header [[n, m], ..."
    try:
        s = line.index('[[') + 2
        return int(line[s:line.index(',', s)])
    except ValueError:
        return parse_line(line)[0]


def collate(items, limit=40000000, hidden_dim=3, batch_replication=1):
    """Loader-batch -> list of segment batches (reference: dag_collate_fn)."""
    edge_nums = [it[2].shape[1] for it in items]
    return [collate_segment([items[j] for j in seg]) for seg in divide(edge_nums, limit // batch_replication, hidden_dim)]


def to_torch(batch, device):
    """numpy batch -> torch tensors in the reference's dtypes, on ``device``."""
    return dict(
        graph_map=torch.from_numpy(batch['graph_map']).to(device),
        batch_variable_map=torch.from_numpy(batch['batch_variable_map']).to(device),
        batch_function_map=torch.from_numpy(batch['batch_function_map']).to(device),
        edge_feature=torch.from_numpy(batch['edge_feature']).to(device),
        label=torch.from_numpy(batch['label']).to(device),
        misc_data=batch['misc_data'], batch_size=batch['batch_size'])


def dimacs_file_list(path):
    """The DIMACS inputs of a run in the converter's order with the converter's labels (reference: src/dimacs2json.py:98-125):
    a directory contributes its *.cnf / *.dimacs files in os.listdir order, label = last digit of the file stem; a single
    file gets the character 8 from the end of its path as label; -1 when that is not a digit."""
    import os
    if os.path.isfile(path):
        c = path[-8] if len(path) >= 8 else ''
        return [(path, float(c) if c.isdigit() else -1)]
    out = []
    for f in os.listdir(path):
        full = os.path.join(path, f)
        stem, ext = os.path.splitext(full)
        if os.path.isfile(full) and ext.lower() in ('.dimacs', '.cnf'):
            out.append((full, float(stem[-1]) if stem[-1].isdigit() else -1))
    return out


def dimacs_item(path, label):
    "One DIMACS file -> loader item, through the native reader (no JSON round trip); same tuple as parse_line of the converted line."
    import os
    from pdp import native
    var_num, clause_num, signed_vars, clause_ids = native.dimacs_parse(path)
    graph_map = np.stack((np.abs(signed_vars) - 1, clause_ids - 1)).astype(np.int32)
    return var_num, clause_num, graph_map, np.sign(signed_vars).astype(np.float32), float(label), [os.path.split(path)[1]]


class FactorGraphDataset(object):
    """JSON-lines dataset with the reference's iteration order.  ``input_file`` may also be a DIMACS file or a
    directory of DIMACS files: the instances are then read directly by the native parser, in the order and with the labels the
    converter (dimacs2json.py) would have produced.  With a ``generator`` (training, dataset.py:84-104) every item is a fresh
    ``generator.generate()`` instance and the data set has ``epoch_size`` items."""

    def __init__(self, input_file, limit, hidden_dim, max_cache_size=100000, batch_replication=1, shard=None, generator=None, epoch_size=0,
                 split_instances=False, split_coupled=False):
        import os
        self._input_file = input_file
        self._dimacs = None
        self._generator = generator
        self._epoch_size = int(epoch_size)
        if generator is not None:
            self._lines = []
        elif os.path.isdir(input_file) or os.path.splitext(input_file)[1].lower() in ('.cnf', '.dimacs'):
            self._dimacs = dimacs_file_list(input_file)
            self._lines = self._dimacs
        else:
            with open(input_file, 'r') as f:
                self._lines = [l for l in f.read().split('\n') if l.strip()]
        # one process per GPU: the loader forms the SAME batches and cuts the SAME segments as a single-process run; this rank collates and
        # yields the segments dealt to it (pdp/parallel.py: one forward = one segment is the reference's coupling domain)
        self._shard = tuple(shard) if shard is not None and (shard[1] > 1 or split_instances) and generator is None else None
        # isolated instances (no coupling inside a segment): every segment is cut into one contiguous instance range per rank instead
        self._split_instances = bool(split_instances) and self._shard is not None and int(batch_replication) == 1
        # the parts stay coupled (--split-forward): every rank takes part in the exchanges of every forward, so every segment needs an instance per rank
        self._split_coupled = bool(split_coupled) and self._split_instances
        self.batch_index = 0          # global index of the loader batch handed out last
        self._limit = limit
        self._hidden_dim = hidden_dim
        self._batch_replication = batch_replication
        self._cache = {}
        self._max_cache_size = max_cache_size

    def __len__(self):
        return self._epoch_size if self._generator is not None else len(self._lines)

    def __getitem__(self, idx):
        if self._generator is not None:
            n, m, graph_map, edge_feature, _, label, _ = self._generator.generate()
            return int(n), int(m), np.asarray(graph_map, dtype=np.int32), np.asarray(edge_feature, dtype=np.float32), float(label), []
        if idx in self._cache:
            return self._cache[idx]
        item = dimacs_item(*self._dimacs[idx]) if self._dimacs is not None else parse_line(self._lines[idx])
        if len(self._cache) < self._max_cache_size:
            self._cache[idx] = item
        return item

    def _edge_count(self, i):
        "edges of instance i, as cheaply as the input format allows (a JSON line is not parsed for it; a DIMACS file is)"
        if i in self._cache:
            return self._cache[i][2].shape[1]
        if self._dimacs is not None:
            return self[i][2].shape[1]
        return json_edge_count(self._lines[i])

    def _variable_count(self, i, tmp=()):
        if i in tmp:
            return tmp[i][0]
        if i in self._cache:
            return self._cache[i][0]
        if self._dimacs is not None:
            return self[i][0]
        return json_variable_count(self._lines[i])

    def _parse_batch(self, idx):
        "DIMACS files: one call parses the not yet cached files of a batch with a few host threads inside the native library"
        import os
        from pdp import native
        todo = [i for i in idx if i not in self._cache]
        tmp = {}
        if len(todo) > 1:
            parsed = native.dimacs_parse_many([self._dimacs[i][0] for i in todo], threads=min(8, os.cpu_count() or 1))
            for i, (vn, cn, sv, ci) in zip(todo, parsed):
                path, label = self._dimacs[i]
                item = (vn, cn, np.stack((np.abs(sv) - 1, ci - 1)).astype(np.int32), np.sign(sv).astype(np.float32), float(label), [os.path.split(path)[1]])
                if len(self._cache) < self._max_cache_size:
                    self._cache[i] = item
                else:
                    tmp[i] = item
        return tmp

    def batches(self, batch_size, order=None):
        """Yields one ``SegmentList`` per loader batch of ``batch_size`` instances (``order``: the sampler's permutation) with the batch's
        global index and the global ids of its segments.  A sharded data set cuts every batch into the same segments as the unsharded
        one (from the edge counts alone), keeps the segments ``parallel.deal_units`` assigns to its rank, and skips batches of which it
        owns nothing."""
        from pdp import parallel
        order = list(range(len(self))) if order is None else list(order)
        loads = [0] * (self._shard[1] if self._shard is not None else 1)
        for j, start in enumerate(range(0, len(order), batch_size)):
            idx = order[start:start + batch_size]
            tmp = self._parse_batch(idx) if self._dimacs is not None else {}
            get = lambda i: tmp[i] if i in tmp else self[i]
            if self._shard is None:
                out = SegmentList(collate([get(i) for i in idx], self._limit, self._hidden_dim, self._batch_replication))
                out.segment_ids = list(range(len(out)))
            else:
                rank, world = self._shard
                edges = [tmp[i][2].shape[1] if i in tmp else self._edge_count(i) for i in idx]
                segments = divide(edges, self._limit // self._batch_replication, self._hidden_dim)
                if self._split_instances:
                    out, ids, parts, whole = SegmentList(), [], [], []
                    for s, seg in enumerate(segments):
                        if self._split_coupled and len(seg) < world:
                            raise ValueError("--split-forward: segment %d of loader batch %d has %d instances for %d ranks (every rank takes part in "
                                             "every forward): use fewer ranks, or --isolated" % (s, j, len(seg), world))
                        lo, hi = parallel.shard_bounds([edges[k] for k in seg], world)[rank]
                        if hi > lo:
                            out.append(collate_segment([get(idx[k]) for k in seg[lo:hi]]))
                            ids.append(s)
                            parts.append((rank, sum(self._variable_count(idx[k], tmp) for k in seg[:lo]), lo))
                            whole.append(lambda seg=seg: collate_segment([get(idx[k]) for k in seg]))
                    if not ids:
                        continue
                    out.segment_ids, out.parts = ids, parts
                    out.whole = whole if self._split_coupled else None
                    out.batch_index = self.batch_index = j
                    yield out
                    continue
                owners = parallel.deal_units([sum(edges[k] for k in seg) for seg in segments], world, loads)
                mine = [s for s, o in enumerate(owners) if o == rank]
                if not mine:
                    continue
                out = SegmentList(collate_segment([get(idx[k]) for k in segments[s]]) for s in mine)
                out.segment_ids = mine
            out.batch_index = self.batch_index = j
            yield out

    @staticmethod
    def get_loader(input_file, limit, hidden_dim, batch_size, shuffle=False, num_workers=0, max_cache_size=100000,
                   use_cuda=True, generator=None, epoch_size=0, batch_replication=1, shard=None, split_instances=False, split_coupled=False):
        """Signature-compatible constructor (reference: dataset.py:189-211); returns an iterable of
        reference-shaped 7-tuples of per-segment lists."""
        ds = FactorGraphDataset(input_file, limit, hidden_dim, max_cache_size, batch_replication, shard=shard, generator=generator, epoch_size=epoch_size,
                                split_instances=split_instances, split_coupled=split_coupled)

        class _Loader(object):
            dataset = ds

            def __iter__(self):
                # torch's DataLoader draws its base seed from the global CPU generator whenever an iterator is
                # created (reference: base.py:258 enumerates the loader once per predict call).  Consume the
                # same draw so that later torch.rand calls see the reference's random stream.
                torch.empty((), dtype=torch.int64).random_()
                order = None
                if shuffle:
                    # torch.utils.data.RandomSampler: a seed from the global generator, then randperm from a generator of its own
                    seed = int(torch.empty((), dtype=torch.int64).random_().item())
                    g = torch.Generator(); g.manual_seed(seed)
                    order = torch.randperm(len(ds), generator=g).tolist()
                for segs in ds.batches(batch_size, order):
                    data = LoaderBatch(([torch.from_numpy(s['graph_map']) for s in segs],
                                        [torch.from_numpy(s['batch_variable_map']) for s in segs],
                                        [torch.from_numpy(s['batch_function_map']) for s in segs],
                                        [torch.from_numpy(s['edge_feature']) for s in segs],
                                        [None for _ in segs],
                                        [torch.from_numpy(s['label']) for s in segs],
                                        [s['misc_data'] for s in segs]))
                    data.index, data.segments, data.parts = segs.batch_index, list(segs.segment_ids), segs.parts
                    data.whole = segs.whole
                    yield data
        return _Loader()

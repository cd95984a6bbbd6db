"""Property tests (hypothesis) of the host-side bookkeeping the multi-GPU path rests on: every rank computes these tables on its own, so they
must be pure functions of their arguments with the stated invariants -- the cut of instances into contiguous ranges (parallel.shard_bounds),
the dealing of units (parallel.deal_units), the dynamic batching policy (dataset.divide, reference: dataset.py:36-72), the counts read off a
JSON line without parsing it, and the Philox keys."""
import json

import numpy as np
from hypothesis import given, settings, strategies as st

from pdp import parallel, generator
from pdp.factorgraph import dataset

EDGES = st.lists(st.integers(min_value=1, max_value=5000), min_size=0, max_size=60)


@settings(max_examples=200, deadline=None, derandomize=True, database=None)
@given(EDGES, st.integers(min_value=1, max_value=9))
def test_shard_bounds_properties(edges, world):
    b = parallel.shard_bounds(edges, world)
    n = len(edges)
    assert len(b) == world and b == parallel.shard_bounds(list(edges), world)                  # pure
    if n == 0:
        assert b == [(0, 0)] * world
        return
    assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(world - 1)) and all(lo <= hi for lo, hi in b)
    nonempty = [hi > lo for lo, hi in b]
    assert sum(nonempty) == min(world, n)                                                       # a rank is idle only when instances ran out
    assert nonempty == sorted(nonempty, reverse=True)                                           # ... and the idle ranks are the last ones
    if n >= world:
        loads = [sum(edges[lo:hi]) for lo, hi in b]
        assert max(loads) <= sum(edges) / world + 2 * max(edges)                                # balanced by edges up to an instance or two


@settings(max_examples=200, deadline=None, derandomize=True, database=None)
@given(st.lists(st.integers(min_value=0, max_value=10 ** 6), min_size=0, max_size=40), st.integers(min_value=1, max_value=8),
       st.lists(st.integers(min_value=0, max_value=10 ** 6), min_size=8, max_size=8))
def test_deal_units_properties(weights, world, carried):
    loads = list(carried[:world])
    before = list(loads)
    owners = parallel.deal_units(weights, world, loads)
    assert len(owners) == len(weights) and all(0 <= o < world for o in owners)
    assert owners == parallel.deal_units(list(weights), world, list(before))                   # pure: every rank computes the same table
    for r in range(world):                                                                      # the carried loads are updated by what was dealt
        assert loads[r] == before[r] + sum(w for w, o in zip(weights, owners) if o == r)
    # longest-processing-time-first onto the least loaded rank: when a rank received its last (= lightest) unit it was the least loaded one, so
    # without that unit it is not above any other rank's final load
    for r in range(world):
        got = [w for w, o in zip(weights, owners) if o == r]
        if got and world > 1:
            assert loads[r] - min(got) <= min(loads[q] for q in range(world) if q != r)


@settings(max_examples=200, deadline=None, derandomize=True, database=None)
@given(st.lists(st.integers(min_value=1, max_value=3000), min_size=1, max_size=50), st.integers(min_value=1, max_value=200000), st.integers(min_value=1, max_value=150))
def test_divide_properties(edges, limit, hidden):
    segs = dataset.divide(edges, limit, hidden)
    flat = [k for seg in segs for k in seg]
    assert sorted(flat) == list(range(len(edges)))                                              # a partition of the batch
    if len(segs) == 1 and flat == list(range(len(edges))):
        return                                                                                  # the whole batch fits: input order (dataset.py:36)
    for seg in segs:
        first = edges[seg[0]]
        assert all(edges[k] <= first for k in seg)                                              # sorted by size, descending
        assert len(seg) <= max(1, limit // (first * hidden))                                    # the reference's allowance, an oversize instance alone
    assert [edges[s[0]] for s in segs] == sorted((edges[s[0]] for s in segs), reverse=True)


@settings(max_examples=100, deadline=None, derandomize=True, database=None)
@given(st.integers(min_value=1, max_value=40), st.integers(min_value=0, max_value=80), st.integers(min_value=0, max_value=2 ** 31 - 1))
def test_counts_read_off_a_json_line(n, m, seed):
    rng = np.random.RandomState(seed)
    k = min(3, n)
    cl = generator.uniform_ksat(n, m, k, rng) if m else []
    line = generator.json_line(n, cl, label=1, name='x')
    vn, fn, gm, ef, label, misc = dataset.parse_line(line)
    assert dataset.json_edge_count(line) == gm.shape[1] and dataset.json_variable_count(line) == vn == json.loads(line)[0][0]


@settings(max_examples=200, deadline=None, derandomize=True, database=None)
@given(st.integers(min_value=0, max_value=2 ** 63), st.integers(min_value=0, max_value=10 ** 6), st.integers(min_value=0, max_value=10 ** 4))
def test_batch_seed_is_a_64_bit_key_and_keeps_the_runs_seed_at_the_origin(seed, j, i):
    key = parallel.batch_seed(seed, j, i)
    assert 0 <= key < 2 ** 64 and parallel.batch_seed(seed, 0, 0) == seed % 2 ** 64
    assert parallel.batch_seed(seed, j, i) == key
    assert parallel.batch_seed(seed, j + 1, i) != key and parallel.batch_seed(seed, j, i + 1) != key     # neighbouring units draw other numbers

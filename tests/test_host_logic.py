"""CPU tests (no GPU): loader layout vs the reference's golden tensors, DIMACS converter vs the reference's output,
dynamic batch divider, C-ABI library loads and exports every declared symbol, product code never touches oracle/."""
import json
import os
import re

import numpy as np
import pytest

from helpers import REPO, load_golden
from pdp import generator
from pdp.factorgraph import dataset


@pytest.mark.parametrize('jsonl,npz', [('mixed_batch.jsonl', 'problem_simplify'), ('sat50_batch.jsonl', 'trace_pdp_n50'),
                                       ('neural_batch.jsonl', 'trace_neural_h32')])
def test_collate_equals_reference_loader(jsonl, npz):
    """dataset.parse_line + collate_segment must produce exactly the tensors the reference loader produced
    (captured in the golden files by running FactorGraphDataset._convert_line / dag_collate_fn)."""
    lines = [l for l in open(os.path.join(REPO, 'tests', 'golden', jsonl)).read().split('\n') if l.strip()]
    b = dataset.collate_segment([dataset.parse_line(l) for l in lines])
    d = load_golden(npz)
    np.testing.assert_array_equal(b['graph_map'], d['graph_map'])
    np.testing.assert_array_equal(b['batch_variable_map'], d['batch_variable_map'])
    np.testing.assert_array_equal(b['batch_function_map'], d['batch_function_map'])
    np.testing.assert_array_equal(b['edge_feature'], d['edge_feature'])
    assert b['graph_map'].dtype == np.int32 and b['edge_feature'].dtype == np.float32 and b['edge_feature'].shape[1] == 1


def test_dimacs_converter_equals_reference(tmp_path):
    import dimacs2json
    out = tmp_path / 'conv.jsonl'
    dimacs2json.convert_directory(os.path.join(REPO, 'tests', 'golden', 'dimacs20'), str(out))
    got = sorted(l for l in out.read_text().split('\n') if l)
    ref = sorted(l for l in open(os.path.join(REPO, 'tests', 'golden', 'cli_dimacs20.converted.jsonl')).read().split('\n') if l)
    assert got == ref
    # file mode: label from the 8th character from the end (reference: dimacs2json.py:118-122)
    src = os.path.join(REPO, 'tests', 'golden', 'dimacs20', 'uf_03_1.cnf')
    dimacs2json.convert_file(src, str(out))
    row = json.loads(out.read_text())
    assert row[4] == ['uf_03_1.cnf'] and row[3] == 0.0      # '...uf_03_1.cnf'[-8] == '0'


def test_converter_edge_cases(tmp_path):
    """duplicate literal (last sign wins), empty clause dropped, unused variables compacted, comment / blank lines"""
    import dimacs2json
    p = tmp_path / 'x_1.cnf'
    p.write_text("c comment\np cnf 6 5\n1 -1 3 0\n\n0\n5 -6 0\n-3 0\nc trailing\n3 5 6 0\n")
    out = tmp_path / 'o.jsonl'
    dimacs2json.convert_directory(str(tmp_path), str(out))
    row = json.loads(out.read_text().strip())
    assert row[0] == [4, 4]                      # variables {1,3,5,6} -> 1..4, the empty clause is gone
    assert row[1] == [-1, 2, 3, -4, -2, 2, 3, 4] and row[2] == [1, 1, 2, 2, 3, 4, 4, 4]
    assert row[3] == 1.0 and row[4] == ['x_1.cnf']


def test_generator_compaction_consistency():
    rng = np.random.RandomState(3)
    variables, signs = generator.uniform_ksat_arrays(30, 100, 3, rng)
    vn, fn, gm, ef = generator.compact_arrays(30, variables, signs)
    clauses = [[int((v + 1) * s) for v, s in zip(vs, ss)] for vs, ss in zip(variables, signs)]
    it = dataset.instance_from_clauses(30, clauses)
    assert (vn, fn) == (it[0], it[1])
    np.testing.assert_array_equal(gm, it[2]); np.testing.assert_array_equal(ef, it[3])


def test_dynamic_batch_divider():
    edges = [30, 10, 50, 20, 40]
    assert dataset.divide(edges, limit=10 ** 9, hidden_dim=3) == [[0, 1, 2, 3, 4]]
    segs = dataset.divide(edges, limit=300, hidden_dim=3)      # 300 // (50*3) = 2 per segment for the largest
    assert segs[0] == [2, 4] and sorted(sum(segs, [])) == [0, 1, 2, 3, 4]
    assert dataset.divide([1000], limit=10, hidden_dim=3) == [[0]]   # larger than the limit: own segment, no infinite loop


def test_c_abi_library_exports_every_declared_symbol():
    """include/pdp_hip.h is the boundary: every function it declares must be exported by libpdp_hip.so."""
    from pdp import native
    header = open(os.path.join(REPO, 'include', 'pdp_hip.h')).read()
    declared = set(re.findall(r'\b(pdp_[a-z0-9_]+)\s*\(', header))
    declared -= {'pdp_solve_args'}
    lib = native.lib()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert set(native.EXPORTED_SYMBOLS) == declared
    assert lib.pdp_abi_version() == 3


def test_no_gpu_means_loud_failure():
    import torch
    from pdp import native
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(native.NativeError):
        native.Problem(torch.zeros(2, 3, dtype=torch.int32), torch.zeros(3, dtype=torch.int32), torch.zeros(1, dtype=torch.int32),
                       torch.ones(3, 1))


def test_product_never_imports_the_oracle():
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, 'pdp-solver_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.h', '.c', '.cpp')) or f == 'Makefile':
                txt = open(os.path.join(root, f), errors='replace').read()
                if re.search(r'(^|\s)(from|import)\s+oracle\b', txt) or 'oracle/' in txt.replace('under oracle/', ''):
                    bad.append(os.path.join(root, f))
    assert not bad, bad


# ---- native DIMACS reader (host code of libpdp_hip.so; no GPU needed) ---------------------------------------------------------------
def _py_compact(path):
    import dimacs2json
    from pdp import generator
    n, clauses = dimacs2json.parse_dimacs(path)
    return generator.compact_instance(n, clauses)


def test_native_dimacs_reader_matches_python_rules(tmp_path):
    from pdp import native
    gdir = os.path.join(REPO, 'tests', 'golden', 'dimacs20')
    files = sorted(os.listdir(gdir))
    assert len(files) == 20
    for f in files:
        vn, cn, sv, ci = native.dimacs_parse(os.path.join(gdir, f))
        pvn, pcn, psv, pci = _py_compact(os.path.join(gdir, f))
        assert (vn, cn) == (pvn, pcn)
        np.testing.assert_array_equal(sv, psv); np.testing.assert_array_equal(ci, pci)
    # the conventions, one by one: comments, '%' lines, duplicate literal, x and -x in one clause (last wins), empty clause,
    # unused variables, a clause without its terminating 0, text after the 0, CRLF and tabs, a sign prefix
    txt = ("c a comment\r\n"
           "p cnf 9 7\r\n"
           "3 -5 3 0\r\n"
           "% not a clause\n"
           "7 -7 2 0\n"
           "0\n"
           "\t-9\t2 0 4 5\n"
           "+3 5\n"
           "\n"
           "-2 -2 -2 0\n")
    p = tmp_path / 'edge.cnf'
    p.write_bytes(txt.encode())
    vn, cn, sv, ci = native.dimacs_parse(str(p))
    pvn, pcn, psv, pci = _py_compact(str(p))
    assert (vn, cn) == (pvn, pcn) == (5, 5)          # variables 2 3 5 7 9 -> 1..5; the "0" line is an empty clause
    np.testing.assert_array_equal(sv, psv); np.testing.assert_array_equal(ci, pci)
    np.testing.assert_array_equal(sv, [2, -3, 1, -4, 1, -5, 2, 3, -1])
    np.testing.assert_array_equal(ci, [1, 1, 2, 2, 3, 3, 4, 4, 5])
    # malformed input is an error with file:line, like the reference's int() failure
    bad = tmp_path / 'bad.cnf'
    bad.write_text("p cnf 2 1\n1 x2 0\n")
    with pytest.raises(native.NativeError, match='bad.cnf:2'):
        native.dimacs_parse(str(bad))
    with pytest.raises(native.NativeError):
        native.dimacs_parse(str(tmp_path / 'missing.cnf'))
    empty = tmp_path / 'empty.cnf'
    empty.write_text("c nothing\np cnf 0 0\n")
    assert native.dimacs_parse(str(empty))[:2] == (0, 0)


def test_dimacs2json_lines_equal_reference_golden(tmp_path):
    "directory conversion through the native reader reproduces the reference converter's lines byte for byte"
    import dimacs2json
    out = tmp_path / 'conv.jsonl'
    dimacs2json.convert_directory(os.path.join(REPO, 'tests', 'golden', 'dimacs20'), str(out))
    got = {json.loads(l)[4][0]: l for l in out.read_text().split('\n') if l.strip()}
    ref = {json.loads(l)[4][0]: l for l in open(os.path.join(REPO, 'tests', 'golden', 'cli_dimacs20.converted.jsonl')).read().split('\n') if l.strip()}
    assert got.keys() == ref.keys() and len(got) == 20
    for k in ref:
        assert got[k] == ref[k], k


def test_loader_reads_dimacs_directly():
    "a DIMACS directory handed to the loader yields exactly the items of the converter's JSON lines (no temp file, no JSON)"
    gdir = os.path.join(REPO, 'tests', 'golden', 'dimacs20')
    ds = dataset.FactorGraphDataset(gdir, 40000000, 3)
    ref = {}
    for l in open(os.path.join(REPO, 'tests', 'golden', 'cli_dimacs20.converted.jsonl')).read().split('\n'):
        if l.strip():
            it = dataset.parse_line(l)
            ref[it[5][0]] = it
    assert len(ds) == 20 == len(ref)
    for i in range(len(ds)):
        vn, fn, gm, ef, label, misc = ds[i]
        r = ref[misc[0]]
        assert (vn, fn, label) == (r[0], r[1], r[4])
        np.testing.assert_array_equal(gm, r[2]); np.testing.assert_array_equal(ef, r[3])
        assert gm.dtype == r[2].dtype and ef.dtype == r[3].dtype
    single = os.path.join(gdir, sorted(os.listdir(gdir))[0])
    one = dataset.FactorGraphDataset(single, 40000000, 3)
    assert len(one) == 1 and one[0][5] == [os.path.split(single)[1]]


def test_native_dimacs_batch_reader(tmp_path):
    "pdp_dimacs_open_many (host threads inside the library) returns what the one-file call returns, in input order; errors propagate"
    from pdp import native
    gdir = os.path.join(REPO, 'tests', 'golden', 'dimacs20')
    paths = [os.path.join(gdir, f) for f in sorted(os.listdir(gdir))]
    many = native.dimacs_parse_many(paths * 3, threads=4)
    assert len(many) == 60
    for path, got in zip(paths * 3, many):
        one = native.dimacs_parse(path)
        assert got[:2] == one[:2]
        np.testing.assert_array_equal(got[2], one[2]); np.testing.assert_array_equal(got[3], one[3])
    assert native.dimacs_parse_many([]) == []
    bad = tmp_path / 'bad.cnf'
    bad.write_text("p cnf 2 1\n1 x2 0\n")
    with pytest.raises(native.NativeError, match='bad.cnf:2'):
        native.dimacs_parse_many(paths[:5] + [str(bad)] + paths[5:9], threads=3)


def test_dataset_deals_instance_ranges_of_isolated_runs(tmp_path):
    """--isolated: the instance is the unit.  Every segment of every loader batch is cut into one contiguous instance range per rank (a
    part); the parts of a segment, in rank order, are the segment of the single-process run, and a part knows the index of its first
    variable / instance inside that segment (where its Philox counters start).  ONE segment keeps every rank busy.  Batch replication
    keeps the segment dealing."""
    from pdp.factorgraph import dataset
    from pdp import generator
    lines = []
    for i in range(23):
        n = 10 + 3 * (i % 5)
        cl = generator.uniform_ksat(n, 3 * n, 3, np.random.RandomState(i))
        lines.append(generator.json_line(n, cl, label=i % 2, name='x%d' % i))
        assert dataset.json_variable_count(lines[-1]) == n
    assert dataset.json_variable_count('[[7,3],[1],[1],0,[]]') == 7
    path = tmp_path / 'in.json'
    path.write_text("\n".join(lines) + "\n")
    ids = lambda seg: [m[0] for m in seg['misc_data']]
    for z, limit in ((23, 10 ** 9), (5, 10 ** 9), (23, 3 * 90 * 2)):
        whole = dataset.FactorGraphDataset(str(path), limit, 3)
        expected = {(segs.batch_index, i): seg for segs in whole.batches(z) for i, seg in zip(segs.segment_ids, segs)}
        for world in (2, 3, 8):
            got = {}
            for rank in range(world):
                ds = dataset.FactorGraphDataset(str(path), limit, 3, shard=(rank, world), split_instances=True)
                for segs in ds.batches(z):
                    assert len(segs) == len(segs.segment_ids) == len(segs.parts) > 0
                    for i, seg, (part, v0, b0) in zip(segs.segment_ids, segs, segs.parts):
                        assert part == rank
                        got.setdefault((segs.batch_index, i), []).append((part, v0, b0, seg))
            assert sorted(got) == sorted(expected)
            for key, parts in got.items():
                exp = expected[key]
                parts.sort(key=lambda t: t[0])
                assert len(parts) == min(world, exp['batch_size'])                        # every rank has a part while instances remain
                assert sum((ids(seg) for _, _, _, seg in parts), []) == ids(exp)
                v_at, b_at = 0, 0
                for part, v0, b0, seg in parts:
                    assert (v0, b0) == (v_at, b_at)
                    nv = seg['batch_variable_map'].size
                    # the part is the slice of the segment: same edges, variable ids shifted by v0
                    sel = (exp['graph_map'][0] >= v0) & (exp['graph_map'][0] < v0 + nv)
                    assert np.array_equal(exp['graph_map'][0][sel] - v0, seg['graph_map'][0])
                    assert np.array_equal(exp['batch_variable_map'][v0:v0 + nv] - b0, seg['batch_variable_map'])
                    v_at += nv; b_at += seg['batch_size']
                assert v_at == exp['batch_variable_map'].size and b_at == exp['batch_size']
    ds = dataset.FactorGraphDataset(str(path), 10 ** 9, 3, batch_replication=2, shard=(0, 2), split_instances=True)
    assert all(segs.parts is None for segs in ds.batches(5))


def test_dataset_deals_segments_to_ranks(tmp_path):
    """one process per GPU: every rank's loader forms the batches of the single-process run, cuts them into the same dynamic segments
    (from the edge counts alone) and collates the segments dealt to it; together the ranks cover every (batch, segment) unit exactly
    once with the content of the single-process run, and ONE loader batch that falls into several segments keeps several ranks busy"""
    from pdp.factorgraph import dataset
    from pdp import generator, parallel
    lines = []
    for i in range(23):
        n = 10 + 3 * (i % 5)
        cl = generator.uniform_ksat(n, 3 * n, 3, np.random.RandomState(i))
        lines.append(generator.json_line(n, cl, label=i % 2, name='x%d' % i))
    path = tmp_path / 'in.json'
    path.write_text("\n".join(lines) + "\n")
    for l in lines:                                                  # the edge count read off the line = the parsed one
        assert dataset.json_edge_count(l) == dataset.parse_line(l)[2].shape[1]
    assert dataset.json_edge_count('[[1, 0], [], [], 0, []]') == 0 and dataset.json_edge_count('[[1,1],[1],[1],0,[]]') == 1
    ids = lambda seg: [m[0] for m in seg['misc_data']]
    for z, limit in ((5, 10 ** 9), (4, 3 * 90 * 2), (23, 3 * 90 * 2)):    # the 2nd / 3rd limit cut every batch into dynamic segments; 3rd: ONE batch
        whole = dataset.FactorGraphDataset(str(path), limit, 3)
        expected = {}
        for segs in whole.batches(z):
            assert segs.segment_ids == list(range(len(segs)))
            for i, seg in zip(segs.segment_ids, segs):
                expected[(segs.batch_index, i)] = (ids(seg), seg['graph_map'].tolist())
        assert sorted({j for j, _ in expected}) == list(range((23 + z - 1) // z))
        assert limit == 10 ** 9 or len(expected) > (23 + z - 1) // z
        for world in (2, 3, 8):
            got, per_rank = {}, []
            for rank in range(world):
                ds = dataset.FactorGraphDataset(str(path), limit, 3, shard=(rank, world))
                mine = 0
                for segs in ds.batches(z):
                    assert len(segs) == len(segs.segment_ids) > 0
                    for i, seg in zip(segs.segment_ids, segs):
                        assert (segs.batch_index, i) not in got
                        got[(segs.batch_index, i)] = (ids(seg), seg['graph_map'].tolist())
                        mine += seg['graph_map'].shape[1]
                per_rank.append(mine)
            assert got == expected
            busy = sum(1 for e in per_rank if e > 0)
            assert busy == min(world, len(expected))                    # also with a single loader batch (z = 23)
            if len(expected) >= 4 * world:
                assert max(per_rank) <= sum(per_rank) / world + max(len(v[1][0]) for v in expected.values())
    # the dealer: heaviest first onto the least loaded rank, loads carried across batches, a pure function
    assert parallel.deal_units([5, 9, 5, 1], 2) == [1, 0, 1, 0]
    loads = [0, 0, 0]
    assert parallel.deal_units([4], 3, loads) == [0] and parallel.deal_units([4], 3, loads) == [1] and parallel.deal_units([4, 4], 3, loads) == [2, 0]
    assert loads == [8, 4, 4]
    assert parallel.deal_units([], 4) == []
    # the loader tuple says where it sits in the run
    loader = dataset.FactorGraphDataset.get_loader(str(path), 3 * 90 * 2, 3, 4, shard=(1, 2))
    seen = [(d.index, d.segments, len(d[0])) for d in loader]
    assert seen and all(len(segs) == n for _, segs, n in seen) and all(isinstance(d, tuple) and len(d) == 7 for d in loader)


def test_cnf_generators_reproduce_the_reference_for_a_seed():
    """uniform / modular / variable-modular generators (reference: src/pdp/generator.py:98-321): same numpy seed, same instance --
    sizes, edge list and signs of two consecutive draws, for generate() and generate_complete()"""
    from pdp import generator as G
    from helpers import load_golden
    d = load_golden('generators')
    specs = [('uniform', lambda: G.UniformCNFGenerator(20, 40, 2, 5, 2.0, 5.0, 5)),
             ('modular', lambda: G.ModularCNFGenerator(3, 30, 60, 0.3, 0.9, 3, 8, 2.0, 5.0, 5)),
             ('vmodular', lambda: G.VariableModularCNFGenerator(2, 5, 30, 60, 0.3, 0.9, 3, 8, 2.0, 5.0, 5))]
    checked = 0
    for name, make in specs:
        for method in ('generate', 'generate_complete'):
            if name == 'vmodular' and method == 'generate_complete':
                with pytest.raises(NotImplementedError):
                    make().generate_complete()
                continue
            for seed in (1, 2, 3):
                np.random.seed(seed)
                g = make()
                for draw in range(2):
                    n, m, gm, ef, _, label, clauses = getattr(g, method)()
                    key = '%s_%s_s%d_d%d' % (name, method, seed, draw)
                    assert [n, m] == d[key + '_nm'].tolist(), key
                    np.testing.assert_array_equal(np.asarray(gm, dtype=np.int32), d[key + '_gm'], err_msg=key)
                    np.testing.assert_array_equal(np.asarray(ef, dtype=np.float32), d[key + '_ef'], err_msg=key)
                    if method == 'generate_complete':
                        assert len(clauses) == m
                    checked += 1
    assert checked == 30


def test_generate_dataset_writes_loadable_files(tmp_path):
    "generate_dataset: JSON lines the loader parses and DIMACS files the native reader parses to the same instance"
    from pdp import generator as G, native
    from pdp.factorgraph import dataset
    np.random.seed(5)
    g = G.ModularCNFGenerator(3, 20, 30, 0.3, 0.9, 3, 6, 2.0, 4.0, alpha_resolution=2)
    g.generate_dataset(3, str(tmp_path / 'dimacs'), str(tmp_path / 'json'), 'mod', sat_only=False)
    jsons = sorted(os.listdir(str(tmp_path / 'json')))
    assert len(jsons) == 2
    lines = [l for l in open(os.path.join(str(tmp_path / 'json'), jsons[0])).read().split('\n') if l.strip()]
    assert len(lines) == 3
    item = dataset.parse_line(lines[0])
    ddir = os.path.join(str(tmp_path / 'dimacs'), jsons[0][:-5])
    files = sorted(os.listdir(ddir))
    assert len(files) == 3
    vn, cn, sv, ci = native.dimacs_parse(os.path.join(ddir, 'dimacs_0_sat=False.DIMACS'))
    assert cn == item[1] and vn == item[0]


def test_converter_subsumption_equals_reference(tmp_path):
    "dimacs2json -s: the two subsumption passes of the reference (dimacs2json.py:60-83), byte-identical lines"
    import dimacs2json
    ddir = os.path.join(REPO, 'tests', 'golden', 'dimacs_subsume')
    out = tmp_path / 's.jsonl'
    dimacs2json.convert_directory(ddir, str(out), True)
    got = {json.loads(l)[4][0]: l for l in out.read_text().split('\n') if l.strip()}
    ref = {json.loads(l)[4][0]: l for l in open(os.path.join(REPO, 'tests', 'golden', 'dimacs_subsume.converted.jsonl')).read().split('\n') if l.strip()}
    plain = {json.loads(l)[4][0]: l for l in open(os.path.join(REPO, 'tests', 'golden', 'dimacs_subsume.plain.jsonl')).read().split('\n') if l.strip()}
    assert got == ref and len(ref) == 4
    assert all(json.loads(ref[k])[0][1] < json.loads(plain[k])[0][1] for k in ref)       # every file really lost clauses


def test_training_loader_follows_torch_dataloader_draws(tmp_path):
    """The training loader (shuffle=True) consumes the global torch CPU generator like torch's DataLoader + RandomSampler do (one base
    seed per iterator, then the sampler's seed for a generator of its own) and visits the items in the sampler's order; with a CNF
    generator every item is a fresh generate() instance and the data set has epoch_size items (reference: dataset.py:84-104, 189-211)."""
    import torch
    import torch.utils.data as tud
    from pdp import generator as gen
    from pdp import cnf_generators
    from pdp.factorgraph.dataset import FactorGraphDataset
    lines = []
    for i in range(11):
        lines.append(gen.json_line(12, gen.uniform_ksat(12, 30, 3, np.random.RandomState(50 + i)), label=i % 2, name='i%d' % i))
    path = tmp_path / 'train.json'
    path.write_text("\n".join(lines) + "\n")

    class Idx(tud.Dataset):
        def __len__(self): return 11
        def __getitem__(self, i): return i

    for epoch_pair in range(2):
        torch.manual_seed(123 + epoch_pair)
        ref_order = []
        for _ in range(2):                                   # two epochs = two iterators over the same DataLoader
            ref_order += [int(x) for batch in tud.DataLoader(Idx(), batch_size=4, shuffle=True, num_workers=0) for x in batch]
        ref_next = torch.rand(3)
        torch.manual_seed(123 + epoch_pair)
        loader = FactorGraphDataset.get_loader(str(path), limit=10 ** 9, hidden_dim=3, batch_size=4, shuffle=True)
        got = []
        for _ in range(2):
            for data in loader:
                got += [int(md[0][1:]) for seg in data[6] for md in seg]
        assert got == ref_order
        assert torch.equal(torch.rand(3), ref_next)
    np.random.seed(5)
    g = cnf_generators.UniformCNFGenerator(6, 12, 2, 4, 2.0, 4.0)
    loader = FactorGraphDataset.get_loader('', limit=10 ** 9, hidden_dim=3, batch_size=4, shuffle=True, generator=g, epoch_size=10)
    sizes = [len(seg) for data in loader for seg in data[6]]
    assert sum(sizes) == 10 and len(loader.dataset) == 10
    np.random.seed(5)
    g2 = cnf_generators.UniformCNFGenerator(6, 12, 2, 4, 2.0, 4.0)
    first = g2.generate()
    np.random.seed(5)
    g3 = cnf_generators.UniformCNFGenerator(6, 12, 2, 4, 2.0, 4.0)
    data = next(iter(FactorGraphDataset.get_loader('', limit=10 ** 9, hidden_dim=3, batch_size=1, shuffle=False, generator=g3, epoch_size=1)))
    assert data[0][0].shape[1] == np.asarray(first[2]).shape[1] and int(data[1][0].numel()) == int(first[0])


def _bench_line_module():
    import sys
    tools = os.path.join(REPO, 'tools')
    if tools not in sys.path:
        sys.path.insert(0, tools)
    from benchlib import line
    return line


def test_bench_final_line_is_compact_strict_json():
    """bench.py's record (tools/benchlib/line.py) built from a canned measurement (tests/golden/bench_measurement.json: a full measurement dict of
    an earlier run): the LAST stdout line holds exactly the contract's keys, is strict JSON (no NaN / Infinity), nests at most two objects deep,
    keeps `config` to <= 25 scalars and stays under 4 KB; every earlier line is a short {"detail": ...} object without the word "metric"."""
    import io
    line = _bench_line_module()
    full = json.load(open(os.path.join(REPO, 'tests', 'golden', 'bench_measurement.json')))
    full['config']['secondary']['neural']['ms_per_iteration_mean'] = float('nan')        # a non-finite figure must not reach the output
    full['roofline']['traffic'] = float('inf')
    buf = io.StringIO()
    last = line.emit(full, out=buf, side_file=False)
    lines = buf.getvalue().split('\n')
    assert lines[-1] == '' and lines[-2] == last

    def strict(s):
        def bad(c):
            raise ValueError(c)
        return json.loads(s, parse_constant=bad)
    rec = strict(last)
    assert len(last.encode()) < 4096
    assert tuple(rec.keys()) == ('metric', 'value', 'unit', 'n_gpus', 'rccl_ranks', 'collective_backend', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                                 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')
    for k in ('config', 'roofline', 'cpu_baseline'):
        assert isinstance(rec[k], dict) and all(not isinstance(v, (dict, list)) for v in rec[k].values()), k
    assert len(rec['config']) <= 25 and rec['config']['workload'].startswith('configs[1]') and rec['config']['E'] == 12600000
    assert tuple(rec['roofline'].keys()) == ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source', 'valu_issue_frac', 'kernel')
    assert rec['roofline']['traffic'] is None and abs(rec['roofline']['frac'] - rec['roofline']['achieved'] / rec['roofline']['peak']) < 1e-5
    assert set(rec['cpu_baseline'].keys()) == {'value', 'unit', 'cores', 'kind', 'cpu_model', 'sample', 'torch_sparse_value', 'torch_sparse_cores'}
    assert abs(rec['value'] - full['value']) < 1e-5 * full['value'] and rec['steps'] == 20 and rec['vs_baseline'] is None
    assert rec['config']['configs2_frac_mfma_f32'] > 0 and rec['config']['solved_T1000_w1000'] == '100/5000'
    details = lines[:-2]
    assert details, 'the per-kernel measurements go to detail lines'
    for l in details:
        d = strict(l)
        assert list(d.keys()) == ['detail', 'data'] and '"metric"' not in l and len(l) < 4096
    names = [strict(l)['detail'] for l in details]
    assert names[-1] == 'summary' and any(n.startswith('config.secondary.neural') for n in names) and 'cpu_baseline_torch_sparse' in ' '.join(names)
    # N > 1 / --no-secondary: no nested measurements, no CPU baseline
    slim = dict(full, cpu_baseline=None, cpu_baseline_torch_sparse=None, summary={})
    slim['config'] = {k: v for k, v in full['config'].items() if not isinstance(v, dict)}
    rec2 = strict(line.emit(slim, out=io.StringIO(), side_file=False))
    assert rec2['cpu_baseline'] is None and 'configs2_it_per_s' not in rec2['config']

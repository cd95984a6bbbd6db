"""Seeded synthetic CNF generators for benchmarks, tests and golden-vector generation.

The reference ships richer generators (reference: src/pdp/generator.py:98-377, out of scope per
SURVEY.md section 2 row 12).  The hot path only needs reproducible *uniform random k-SAT* inputs of
the shape BASELINE.json names (n=200, m=840, k=3 ...), so this module provides exactly that plus
the DIMACS writer used by the CLI tests.  Everything is driven by ``numpy.random.RandomState`` so
the same (seed, n, m, k) always yields the same instance on every machine.
"""

import os

import numpy as np


def uniform_ksat(n, m, k, rng):
    """Return ``m`` clauses over ``n`` variables, each with ``k`` distinct variables drawn uniformly
    and independent fair signs.  Clauses are lists of non-zero signed 1-based ints (DIMACS style)."""
    clauses = []
    for _ in range(m):
        variables = rng.choice(n, size=k, replace=False) + 1
        signs = rng.randint(0, 2, size=k) * 2 - 1
        clauses.append([int(v * s) for v, s in zip(variables, signs)])
    return clauses


def uniform_ksat_arrays(n, m, k, rng):
    """Vectorised uniform k-SAT: returns (variables int32 [m,k] 0-based, ascending within a clause;
    signs int8 [m,k] in {-1,+1}).  Same distribution as ``uniform_ksat`` (distinct variables per
    clause, fair signs) but a different random stream; used for the large benchmark batches."""
    variables = rng.randint(0, n, size=(m, k))
    while True:
        srt = np.sort(variables, axis=1)
        dup = (srt[:, 1:] == srt[:, :-1]).any(axis=1)
        if not dup.any():
            break
        variables[dup] = rng.randint(0, n, size=(int(dup.sum()), k))
    signs = (rng.randint(0, 2, size=(m, k)) * 2 - 1).astype(np.int8)
    order = np.argsort(variables, axis=1, kind='stable')
    variables = np.take_along_axis(variables, order, axis=1).astype(np.int32)
    signs = np.take_along_axis(signs, order, axis=1)
    return variables, signs


def compact_arrays(n, variables, signs):
    """Array form of ``compact_instance`` for fixed-k clause matrices with distinct variables per clause:
    drops unused variables (ascending renumbering).  Returns (var_num, clause_num, graph_map int32 [2,E],
    edge_feature float32 [E]) in the loader's clause-major order."""
    m, k = variables.shape
    used = np.unique(variables)
    if used.size == n:
        local = variables
    else:
        remap = np.full(n, -1, dtype=np.int32)
        remap[used] = np.arange(used.size, dtype=np.int32)
        local = remap[variables]
    graph_map = np.stack((local.reshape(-1), np.repeat(np.arange(m, dtype=np.int32), k))).astype(np.int32)
    return int(used.size), int(m), graph_map, signs.reshape(-1).astype(np.float32)


def clause_count(n, k, alpha=None):
    """Number of clauses for the benchmark family: threshold-ish ratios."""
    if alpha is None:
        alpha = {3: 4.2, 4: 0.9 * 9.93, 5: 0.9 * 21.12}.get(k, 4.2)
    return int(round(alpha * n))


def write_dimacs(path, n, clauses):
    """Write the single-space / ' 0'-terminated DIMACS dialect the reference parser accepts
    (reference: src/dimacs2json.py:30-45)."""
    with open(path, 'w') as f:
        f.write("p cnf %d %d\n" % (n, len(clauses)))
        for c in clauses:
            f.write(" ".join(str(l) for l in c) + " 0\n")


def compact_instance(n, clauses):
    """Canonicalise an instance the way the reference's DIMACS->JSON converter does
    (reference: src/dimacs2json.py:43-51,85-91): within a clause the last occurrence of a variable
    wins, empty clauses are dropped, unused variables are removed (remaining ones renumbered in
    ascending order), literals are listed clause-major with ascending variable index.

    Returns (var_num, clause_num, signed_vars int32[E], clause_ids int32[E]) with 1-based ids."""
    rows = []
    for c in clauses:
        lit = {}
        for l in c:
            l = int(l)
            if l == 0:
                continue
            lit[abs(l)] = 1 if l > 0 else -1
        if lit:
            rows.append(lit)
    used = sorted({v for r in rows for v in r})
    remap = {v: i + 1 for i, v in enumerate(used)}
    signed_vars, clause_ids = [], []
    for ci, r in enumerate(rows):
        for v in sorted(r):
            signed_vars.append(remap[v] * r[v])
            clause_ids.append(ci + 1)
    return (len(used), len(rows), np.asarray(signed_vars, dtype=np.int32),
            np.asarray(clause_ids, dtype=np.int32))


def json_line(n, clauses, label=-1, name=""):
    """One line of the compact JSON dataset format (reference: src/dimacs2json.py:85-91,
    src/pdp/factorgraph/dataset.py:120-136)."""
    var_num, clause_num, sv, ci = compact_instance(n, clauses)
    return format_json_line(var_num, clause_num, sv, ci, label, name)


def format_json_line(var_num, clause_num, signed_vars, clause_ids, label=-1, name=""):
    "The JSON text of one compact instance (same byte layout as json.dumps of the reference's list, dimacs2json.py:85-91)."
    label_txt = repr(float(label)) if isinstance(label, float) else str(label)
    return "[[%d, %d], [%s], [%s], %s, [\"%s\"]]" % (
        var_num, clause_num, ", ".join(map(str, np.asarray(signed_vars).tolist())),
        ", ".join(map(str, np.asarray(clause_ids).tolist())), label_txt, name)


def generate_batch(batch, n, k=3, m=None, seed=0):
    """``batch`` independent instances; instance ``i`` uses RandomState(seed + i)."""
    if m is None:
        m = clause_count(n, k)
    return [(n, uniform_ksat(n, m, k, np.random.RandomState(seed + i))) for i in range(batch)]


def write_dimacs_directory(directory, instances, stem="inst"):
    os.makedirs(directory, exist_ok=True)
    paths = []
    for i, (n, clauses) in enumerate(instances):
        p = os.path.join(directory, "%s_%04d.cnf" % (stem, i))
        write_dimacs(p, n, clauses)
        paths.append(p)
    return paths


# the generator classes of the reference's pdp.generator module (uniform and the two Community Attachment variants)
try:
    from pdp.cnf_generators import (CNFGeneratorBase, UniformCNFGenerator, ModularCNFGenerator,  # noqa: E402,F401
                                    VariableModularCNFGenerator, is_sat)
except ImportError:      # this file loaded on its own next to another `pdp` package (tests/golden/generate_golden.py does that)
    pass

#!/usr/bin/env python3
"""DIMACS -> compact JSON converter (drop-in for the reference's src/dimacs2json.py).

Output lines are byte-identical to the reference's for valid input (``[[n, m], [signed vars], [clause ids], label,
[file name]]``, reference: dimacs2json.py:85-91,111,125) including its conventions: the last occurrence of a
variable inside a clause wins, empty clauses and unused variables are dropped, literals are clause-major with
ascending variable index, the label is the last digit of the file stem (directory mode, :105) or the character
8 from the end of the path (file mode, :118-122).  The parser streams clauses into sparse rows instead of the
reference's dense [clauses x variables] matrix (native single-pass parser, csrc/pdp_dimacs.hip), so big instances do
not need O(n*m) memory.  ``-s`` removes subsumed clauses in the reference's two passes (dimacs2json.py:60-83) through an inverted
literal index instead of its dense [clauses x clauses] product.
"""

import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from pdp import generator  # noqa: E402


def parse_dimacs(path):
    """Pure-Python statement of the parsing rules: (declared variable count, list of clauses as lists of signed ints).
    The converter itself uses the native parser; the tests check the two against each other."""
    n = 0
    clauses = []
    with open(path, 'r') as f:
        for line in f:
            tok = line.split()
            if not tok or tok[0] == 'c' or tok[0] == '%':
                continue
            if tok[0] == 'p':
                n = int(tok[2])
                continue
            lits = []
            for t in tok:
                v = int(t)
                if v == 0:
                    break
                lits.append(v)
            clauses.append(lits)
    return n, clauses


def remove_subsumed(signed_vars, clause_ids):
    """The reference's ``_propagate_constraints`` (dimacs2json.py:60-83) on the compact edge list: pass 1 drops every clause that contains
    all literals of a LATER clause (equal clauses: the earlier one goes), pass 2 drops, among the survivors, every clause that contains
    all literals of an EARLIER one.  Variables keep their numbers (the reference compacts them before this step only).  Returns
    (signed_vars, clause_ids) with the surviving clauses renumbered 1..m' in order."""
    import numpy as np
    sv = np.asarray(signed_vars, dtype=np.int64); ci = np.asarray(clause_ids, dtype=np.int64)
    m = int(ci.max()) if ci.size else 0
    bounds = np.searchsorted(ci, np.arange(1, m + 2))
    clauses = [sv[bounds[i]:bounds[i + 1]].tolist() for i in range(m)]

    def survivors(cls, drop_superset_of_later):
        index = {}
        for i, c in enumerate(cls):
            for lit in c:
                index.setdefault(lit, []).append(i)
        keep = [True] * len(cls)
        for i, c in enumerate(cls):                        # clause i as the (candidate) subset
            if not c:
                continue
            lists = sorted((index[lit] for lit in c), key=len)
            common = set(lists[0])
            for lst in lists[1:]:
                common.intersection_update(lst)
                if not common:
                    break
            for j in common:                               # clause j contains every literal of clause i
                if drop_superset_of_later and j < i:
                    keep[j] = False
                if not drop_superset_of_later and j > i:
                    keep[j] = False
        return [c for c, k in zip(cls, keep) if k]

    if len(clauses) >= 2:
        clauses = survivors(clauses, True)
    if len(clauses) >= 2:
        clauses = survivors(clauses, False)
    out_sv = [lit for c in clauses for lit in c]
    out_ci = [i + 1 for i, c in enumerate(clauses) for _ in c]
    return np.asarray(out_sv, dtype=np.int32), np.asarray(out_ci, dtype=np.int32)


def json_line(path, label, propagate=False):
    "One output line; the text is read by the native parser of libpdp_hip.so (pdp_dimacs_open, include/pdp_hip.h)."
    from pdp import native
    var_num, clause_num, signed_vars, clause_ids = native.dimacs_parse(path)
    if propagate:
        signed_vars, clause_ids = remove_subsumed(signed_vars, clause_ids)
        clause_num = int(clause_ids.max()) if clause_ids.size else 0
    return generator.format_json_line(var_num, clause_num, signed_vars, clause_ids, label=label, name=os.path.split(path)[1])


def convert_directory(dimacs_dir, output_file, propagate=False, only_positive=False):
    file_list = [os.path.join(dimacs_dir, f) for f in os.listdir(dimacs_dir) if os.path.isfile(os.path.join(dimacs_dir, f))]
    with open(output_file, 'w') as f:
        for path in file_list:
            name, ext = os.path.splitext(path)
            if ext.lower() not in ('.dimacs', '.cnf'):
                continue
            label = float(name[-1]) if name[-1].isdigit() else -1
            if only_positive and label == 0:
                continue
            f.write(json_line(path, label, propagate) + '\n')


def convert_file(file_name, output_file, propagate=False):
    if len(file_name) < 8:
        label = -1
    else:
        c = file_name[-8]
        label = float(c) if c.isdigit() else -1
    with open(output_file, 'w') as f:
        f.write(json_line(file_name, label, propagate) + '\n')


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('in_dir', action='store', type=str)
    parser.add_argument('out_file', action='store', type=str)
    parser.add_argument('-s', '--simplify', help='Propagate binary constraints', required=False, action='store_true', default=False)
    parser.add_argument('-p', '--positive', help='Output only positive examples', required=False, action='store_true', default=False)
    args = vars(parser.parse_args())
    convert_directory(args['in_dir'], args['out_file'], args['simplify'], args['positive'])

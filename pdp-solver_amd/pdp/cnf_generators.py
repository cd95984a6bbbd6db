"""Random CNF generators of the PDP framework: uniform k-SAT and the two Community Attachment variants
(reference: src/pdp/generator.py:22-377; model: Giraldez-Cru & Levy, "Generating SAT instances with community structure").

Host-side numpy code for data-set creation (the reference also feeds its training loop with them, which is out of scope here).
Every generator draws from numpy's global generator in the reference's order, so ``np.random.seed(s)`` followed by ``generate()`` /
``generate_complete()`` returns the reference's instance (tests/golden/generators.npz).  ``generate`` is the fast unlabeled form,
``generate_complete`` rejects duplicate clauses (ten trials per clause) and labels the instance through ``is_sat``.
"""

import os
import sys

import numpy as np


def is_sat(_var_num, iclause_list):
    "Hook for a SAT solver that labels generated instances (the reference ships the same stub, generator.py:15-17)."
    return False


class _Communities(object):
    "n variables cut into c consecutive blocks of a random permutation; the last block takes the remainder"

    def __init__(self, n, c, q):
        self.n, self.c = n, c
        self.size = int(n / c)
        self.sizes = self.size * np.ones(c, dtype=np.int32)
        self.sizes[c - 1] += (n - np.sum(self.sizes))
        self.p_same = q + 1.0 / c
        self.index = np.random.permutation(n)

    def draw(self, length, fallback_to_uniform):
        "variables of one clause: all from one community with probability p_same, otherwise from `length` different ones"
        if np.random.uniform() <= self.p_same:
            community = np.random.randint(0, self.c)
            start = self.size * community
            return self.index[np.random.choice(np.arange(start, start + self.sizes[community]), length, replace=False)]
        if not fallback_to_uniform or self.c >= length:
            communities = np.random.choice(self.c, length, replace=False)
            inner = (np.random.uniform(size=length) * self.sizes[communities]).astype(int)
            return self.index[self.size * communities + inner]
        return np.random.choice(self.n, length, replace=False)


class CNFGeneratorBase(object):
    "Common part: the (n, alpha) ranges, the alpha sweep of ``generate_dataset`` and the two file formats."

    def __init__(self, min_n, max_n, min_alpha, max_alpha, alpha_resolution=10):
        self._min_n, self._max_n = min_n, max_n
        self._min_alpha, self._max_alpha = min_alpha, max_alpha
        self._alpha = min_alpha
        self._alpha_inc = (max_alpha - min_alpha) / alpha_resolution
        self._alpha_resolution = alpha_resolution

    def generate(self):
        raise NotImplementedError

    def generate_complete(self):
        raise NotImplementedError

    # -- shared building blocks ---------------------------------------------------------------------------------------------
    def _draw_size(self, alpha_lo, alpha_hi):
        n = np.random.randint(self._min_n, self._max_n + 1)
        alpha = np.random.uniform(alpha_lo, alpha_hi)
        return n, int(n * alpha)

    @staticmethod
    def _assemble(clause_variables):
        "clause-major edge list from per-clause variable arrays"
        lengths = [len(v) for v in clause_variables]
        graph_map = np.zeros((2, int(np.sum(lengths))), dtype=np.int32)
        pos = 0
        for i, v in enumerate(clause_variables):
            graph_map[0, pos:pos + len(v)] = v
            graph_map[1, pos:pos + len(v)] = i
            pos += len(v)
        return graph_map

    def _complete(self, n, m, draw_clause):
        """m clauses with duplicate rejection (the clause index only advances when an unseen clause was found within ten trials;
        an exhausted clause is kept under the previous index, like the reference)."""
        seen, clause_list = set(), []
        graph_map = np.zeros((2, 0), dtype=np.int32)
        edge_features = np.zeros(0)
        i = -1
        for _ in range(m):
            for _ in range(10):
                literals = np.sort(draw_clause())
                signs = 2.0 * np.random.choice(2, len(literals)) - 1
                iclause = [int(x) for x in ((literals + 1) * signs).astype(int)]
                if str(iclause) not in seen:
                    i += 1
                    break
            seen.add(str(iclause))
            clause_list.append(iclause)
            graph_map = np.concatenate((graph_map, np.stack((literals, i * np.ones(len(literals), dtype=np.int32)))), 1)
            edge_features = np.concatenate((edge_features, signs))
        return n, m, graph_map, edge_features, None, is_sat(n, clause_list), clause_list

    # -- file formats (generator.py:42-51) ------------------------------------------------------------------------------------
    @staticmethod
    def _to_json(n, m, graph_map, edge_feature, label):
        # (a bool label is written as 0 / 1: the reference's str(False) is not JSON and its own loader cannot read it back)
        return [[int(n), int(m)], [int(x) for x in ((graph_map[0, :] + 1) * edge_feature).astype(int)],
                [int(x) for x in (graph_map[1, :] + 1)], int(label) if isinstance(label, (bool, np.bool_)) else label]

    @staticmethod
    def _to_dimacs(n, m, clause_list):
        return 'p cnf %d %d\n' % (n, m) + ''.join(' '.join(str(int(l)) for l in clause) + ' 0\n' for clause in clause_list)

    def generate_dataset(self, size, output_dimacs_path, json_output, name, sat_only=True):
        "``alpha_resolution`` JSON files + DIMACS directories, one per alpha slice (generator.py:53-93)"
        os.makedirs(output_dimacs_path, exist_ok=True)
        os.makedirs(json_output, exist_ok=True)
        dimacs_base, json_base = os.path.join(output_dimacs_path, name), os.path.join(json_output, name)
        for j in range(self._alpha_resolution):
            postfix = '_%d_%s_%s' % (j, self._alpha, self._alpha + self._alpha_inc)
            os.makedirs(dimacs_base + postfix, exist_ok=True)
            with open(json_base + postfix + ".json", 'w') as f:
                for i in range(size):
                    found = False
                    for _ in range(50):
                        n, m, graph_map, edge_feature, _, label, clause_list = self.generate_complete()
                        if (not sat_only) or (label == 1):
                            found = True
                            break
                    if found:
                        f.write(str(self._to_json(n, m, graph_map, edge_feature, label)).replace("'", '"') + '\n')
                        with open(os.path.join(dimacs_base + postfix, 'dimacs_%d_sat=%s.DIMACS' % (i, label)), 'w') as g:
                            g.write(self._to_dimacs(n, m, clause_list) + '\n')
                    sys.stdout.write("Dataset {:2d}/{:2d}: {:.2f} % complete  \r".format(j + 1, self._alpha_resolution, 100 * float(i + 1) / size))
                    sys.stdout.flush()
            self._alpha += self._alpha_inc


class UniformCNFGenerator(CNFGeneratorBase):
    "Uniformly random CNF with clause lengths in [min_k, max_k] (generator.py:98-160)."

    def __init__(self, min_n, max_n, min_k, max_k, min_alpha, max_alpha, alpha_resolution=10):
        super(UniformCNFGenerator, self).__init__(min_n, max_n, min_alpha, max_alpha, alpha_resolution)
        self._min_k, self._max_k = min_k, max_k

    def _length(self, n):
        return np.random.randint(self._min_k, min(self._max_k, n - 1) + 1)

    def generate(self):
        n, m = self._draw_size(self._min_alpha, self._max_alpha)
        lengths = [self._length(n) for _ in range(m)]
        graph_map = self._assemble([np.random.choice(n, k, replace=False) for k in lengths])
        edge_feature = 2.0 * np.random.choice(2, graph_map.shape[1]) - 1
        return n, m, graph_map, edge_feature, None, -1.0, []

    def generate_complete(self):
        n, m = self._draw_size(self._alpha, self._alpha + self._alpha_inc)
        return self._complete(n, m, lambda: np.random.choice(n, self._length(n), replace=False))


class ModularCNFGenerator(CNFGeneratorBase):
    "Community Attachment model with clauses of fixed length k (generator.py:163-265)."

    def __init__(self, k, min_n, max_n, min_q, max_q, min_c, max_c, min_alpha, max_alpha, alpha_resolution=10):
        super(ModularCNFGenerator, self).__init__(min_n, max_n, min_alpha, max_alpha, alpha_resolution)
        self._k = k
        self._min_c, self._max_c, self._min_q, self._max_q = min_c, max_c, min_q, max_q

    def _communities(self, n, least):
        q = np.random.uniform(self._min_q, self._max_q)
        c = np.random.randint(self._min_c, self._max_c + 1)
        return _Communities(n, max(least, min(c, int(n / self._k) - 1)), q)

    def generate(self):
        n, m = self._draw_size(self._min_alpha, self._max_alpha)
        com = self._communities(n, 1)
        graph_map = self._assemble([com.draw(self._k, True) for _ in range(m)])
        edge_feature = 2.0 * np.random.choice(2, m * self._k) - 1
        return n, m, graph_map, edge_feature, None, -1.0, []

    def generate_complete(self):
        n, m = self._draw_size(self._alpha, self._alpha + self._alpha_inc)
        com = self._communities(n, self._k + 1)
        return self._complete(n, m, lambda: com.draw(self._k, False))


class VariableModularCNFGenerator(CNFGeneratorBase):
    "Community Attachment model with clause lengths in [min_k, max_k] (generator.py:270-321)."

    def __init__(self, min_k, max_k, min_n, max_n, min_q, max_q, min_c, max_c, min_alpha, max_alpha, alpha_resolution=10):
        super(VariableModularCNFGenerator, self).__init__(min_n, max_n, min_alpha, max_alpha, alpha_resolution)
        self._min_k, self._max_k = min_k, max_k
        self._min_c, self._max_c, self._min_q, self._max_q = min_c, max_c, min_q, max_q

    def generate(self):
        n, m = self._draw_size(self._min_alpha, self._max_alpha)
        q = np.random.uniform(self._min_q, self._max_q)
        c = max(1, min(np.random.randint(self._min_c, self._max_c + 1), n))
        size = int(n / c)
        lengths = [np.random.randint(min(self._min_k, size), min(self._max_k, n - 1, size) + 1) for _ in range(m)]
        com = _Communities(n, c, q)
        graph_map = self._assemble([com.draw(k, True) for k in lengths])
        edge_feature = 2.0 * np.random.choice(2, graph_map.shape[1]) - 1
        return n, m, graph_map, edge_feature, None, -1.0, []

    def generate_complete(self):
        raise NotImplementedError("VariableModularCNFGenerator.generate_complete reads an attribute the reference never sets "
                                  "(generator.py:332, SURVEY.md App. B-11): it cannot run there either")

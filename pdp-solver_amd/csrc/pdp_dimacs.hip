// pdp_dimacs.hip -- native DIMACS CNF reader (host code only; compiled into libpdp_hip.so next to the kernels).
//
// Replaces the text side of the reference's converter for the inference path (reference: src/dimacs2json.py:22-51 parsing,
// :43-51,85-91 compaction) with one streaming pass: the reference fills a dense [clauses x variables] numpy matrix per file,
// this reader produces the compact edge list the loader / pdp_problem_create consume, in the same conventions:
//   * a line whose first token is "c" or "%" is a comment; "p cnf <vars> <clauses>" is read and otherwise ignored;
//   * every other line is ONE clause: literals up to the first 0 (or the end of the line);
//   * inside a clause the LAST occurrence of a variable wins; empty clauses are dropped;
//   * unused variables are removed, the remaining ones renumbered in ascending order;
//   * edges are clause-major with ascending variable index inside a clause; ids are 1-based, the literal sign is the sign
//     of the variable id (exactly the second and third list of the reference's JSON line).
#include "pdp_common.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <thread>
#include <atomic>
#include <string>

struct pdp_dimacs {
    int32_t n_vars = 0, n_clauses = 0;
    std::vector<int32_t> signed_vars, clause_ids;
};

static inline bool is_space(unsigned char c) { return c == ' ' || c == '\t' || c == '\f' || c == '\v'; }
static inline bool is_eol(unsigned char c) { return c == '\n' || c == '\r'; }

extern "C" int pdp_dimacs_open(const char *path, pdp_dimacs **out, int32_t *n_vars, int32_t *n_clauses, int64_t *n_edges)
{
    PDP_REQUIRE(path && out && n_vars && n_clauses && n_edges, "NULL argument");
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) { pdp_set_error("cannot open %s", path); return PDP_ERR_INVALID; }
    std::vector<char> buf;
    {
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        buf.resize(sz > 0 ? (size_t)sz : 0);
        const size_t got = buf.empty() ? 0 : fread(buf.data(), 1, buf.size(), f);
        fclose(f);
        if (got != buf.size()) { pdp_set_error("short read on %s", path); return PDP_ERR_INVALID; }
    }
    pdp_dimacs *d = new pdp_dimacs();
    std::vector<int32_t> lits;            // literals of all non-empty clauses, already de-duplicated and sorted per clause
    std::vector<int64_t> clause_start;    // offsets into lits
    std::vector<int32_t> cur;
    int32_t max_var = 0;
    const char *p = buf.data(), *end = p + buf.size();
    long line_no = 0;
    while (p < end) {
        // one line
        ++line_no;
        const char *eol = p;
        while (eol < end && !is_eol((unsigned char)*eol)) ++eol;
        const char *q = p;
        while (q < eol && is_space((unsigned char)*q)) ++q;
        bool is_clause = q < eol;
        if (is_clause) {
            const char *t = q;
            while (t < eol && !is_space((unsigned char)*t)) ++t;
            const size_t len = (size_t)(t - q);
            if (len == 1 && (*q == 'c' || *q == '%')) is_clause = false;
            else if (len == 1 && *q == 'p') {
                // "p cnf <vars> ...": the third token must be an integer (the reference reads it), its value is not needed
                int tokens = 0; const char *s = q; bool ok = false;
                while (s < eol) {
                    while (s < eol && is_space((unsigned char)*s)) ++s;
                    if (s >= eol) break;
                    const char *e = s;
                    while (e < eol && !is_space((unsigned char)*e)) ++e;
                    if (++tokens == 3) { char *ep = nullptr; (void)strtol(s, &ep, 10); ok = (ep == e); break; }
                    s = e;
                }
                if (!ok) { pdp_set_error("%s:%ld: malformed problem line", path, line_no); delete d; return PDP_ERR_INVALID; }
                is_clause = false;
            }
        }
        if (is_clause) {
            cur.clear();
            const char *s = q;
            while (s < eol) {
                while (s < eol && is_space((unsigned char)*s)) ++s;
                if (s >= eol) break;
                const char *e = s;
                bool neg = false;
                if (*e == '-' || *e == '+') { neg = (*e == '-'); ++e; }
                int64_t v = 0; int digits = 0;
                while (e < eol && *e >= '0' && *e <= '9') { v = v * 10 + (*e - '0'); ++e; ++digits; if (v > 0x7fffffff) break; }
                if (digits == 0 || (e < eol && !is_space((unsigned char)*e)) || v > 0x7fffffff) {
                    pdp_set_error("%s:%ld: not an integer literal", path, line_no); delete d; return PDP_ERR_INVALID;
                }
                if (v == 0) break;                          // clause terminator: the rest of the line is ignored
                cur.push_back(neg ? -(int32_t)v : (int32_t)v);
                s = e;
            }
            if (!cur.empty()) {
                // last occurrence of a variable wins; ascending variable index
                std::stable_sort(cur.begin(), cur.end(), [](int32_t a, int32_t b) { return std::abs(a) < std::abs(b); });
                clause_start.push_back((int64_t)lits.size());
                for (size_t i = 0; i < cur.size(); ++i) {
                    if (i + 1 < cur.size() && std::abs(cur[i + 1]) == std::abs(cur[i])) continue;
                    lits.push_back(cur[i]);
                    if (std::abs(cur[i]) > max_var) max_var = std::abs(cur[i]);
                }
            }
        }
        p = eol;
        if (p < end && *p == '\r') ++p;
        if (p < end && *p == '\n') ++p;
    }
    clause_start.push_back((int64_t)lits.size());
    // drop unused variables (ascending renumbering)
    std::vector<int32_t> remap((size_t)max_var + 1, 0);
    for (int32_t l : lits) remap[(size_t)std::abs(l)] = 1;
    int32_t next = 0;
    for (size_t v = 1; v < remap.size(); ++v) if (remap[v]) remap[v] = ++next;
    d->n_vars = next;
    d->n_clauses = (int32_t)(clause_start.size() - 1);
    d->signed_vars.resize(lits.size()); d->clause_ids.resize(lits.size());
    for (int32_t c = 0; c < d->n_clauses; ++c)
        for (int64_t k = clause_start[(size_t)c]; k < clause_start[(size_t)c + 1]; ++k) {
            const int32_t l = lits[(size_t)k];
            d->signed_vars[(size_t)k] = l > 0 ? remap[(size_t)l] : -remap[(size_t)-l];
            d->clause_ids[(size_t)k] = c + 1;
        }
    *out = d; *n_vars = d->n_vars; *n_clauses = d->n_clauses; *n_edges = (int64_t)lits.size();
    return PDP_OK;
}

extern "C" int pdp_dimacs_read(const pdp_dimacs *d, int32_t *signed_vars, int32_t *clause_ids)
{
    PDP_REQUIRE(d && (d->signed_vars.empty() || (signed_vars && clause_ids)), "NULL argument");
    std::copy(d->signed_vars.begin(), d->signed_vars.end(), signed_vars);
    std::copy(d->clause_ids.begin(), d->clause_ids.end(), clause_ids);
    return PDP_OK;
}

extern "C" int pdp_dimacs_close(pdp_dimacs *d)
{
    delete d;
    return PDP_OK;
}

// Many files at once, parsed by a few host threads (the loader's DIMACS mode: thousands of small files per batch).
// out[i] receives the handle of paths[i] (NULL for a file that failed; the first failure's message is kept), sizes as in pdp_dimacs_open.
extern "C" int pdp_dimacs_open_many(const char *const *paths, int32_t count, int32_t threads, pdp_dimacs **out, int32_t *n_vars,
                                    int32_t *n_clauses, int64_t *n_edges)
{
    PDP_REQUIRE(count >= 0 && (count == 0 || (paths && out && n_vars && n_clauses && n_edges)), "NULL argument");
    if (threads < 1) threads = 1;
    if (threads > count) threads = count > 0 ? count : 1;
    std::atomic<int32_t> next(0), failed(-1);
    std::vector<std::string> messages((size_t)threads);
    auto work = [&](int tid) {
        for (;;) {
            const int32_t i = next.fetch_add(1);
            if (i >= count) break;
            out[i] = nullptr;
            const int st = pdp_dimacs_open(paths[i], &out[i], &n_vars[i], &n_clauses[i], &n_edges[i]);
            if (st != PDP_OK) {
                int32_t expect = -1;
                if (failed.compare_exchange_strong(expect, i)) messages[(size_t)tid] = pdp_last_error();
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
    work(0);
    for (std::thread &t : pool) t.join();
    if (failed.load() >= 0) {
        for (const std::string &m : messages) if (!m.empty()) { pdp_set_error("%s", m.c_str()); break; }
        for (int32_t i = 0; i < count; ++i) if (out[i]) { delete out[i]; out[i] = nullptr; }
        return PDP_ERR_INVALID;
    }
    return PDP_OK;
}

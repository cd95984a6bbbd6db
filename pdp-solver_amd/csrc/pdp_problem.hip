// pdp_problem.hip -- batch container: builds the HBM-resident, instance-local layout of a batch of
// CNF instances.  replaces: SATProblem.__init__/setup_problem + mask builders + _replicate_batch
// (reference: src/pdp/nn/solver.py:22-178).
//
// HBM layout produced here (all int32 unless noted; b = instance, ids local to the instance):
//   inst_v0/f0/e0 [B+1]  first variable / clause / edge of each instance
//   e_var, e_fn   [E]    local variable / clause of each edge      e_sgn [E] int8 literal sign
//   v_ptr  [V+B]         CSR offsets of instance b at v0+b (n+1 entries), v_edges [E] local edge ids
//   f_ptr  [F+B]         same by clause,                           f_edges [E]
// Rows keep ascending edge id, which pins the fp32 summation order to the reference's
// (torch.mm(sparse, dense) on CPU accumulates in storage order).
#include "pdp_common.hpp"
#include <mutex>
#include <unordered_map>
#include <vector>

#include <hipcub/hipcub.hpp>
#include <stdarg.h>

static thread_local char g_err[512] = "";

void pdp_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *pdp_last_error(void) { return g_err; }
extern "C" int pdp_abi_version(void) { return PDP_ABI_VERSION; }
extern "C" int pdp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- kernel timing (measurement only) -------------------------------------------------------------------------------------
int g_pdp_timing_on = 0;
const char *g_pdp_kernel_name[PDP_KN_COUNT] = {nullptr};
namespace {
struct TimedSpan { int key; hipEvent_t e0, e1; bool closed; };
std::mutex g_timing_mu;
std::vector<TimedSpan> g_spans;
std::vector<hipEvent_t> g_event_pool;
hipEvent_t timing_event()
{
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}
void pdp_timing_mark(int key, hipStream_t st, bool begin)
{
    std::lock_guard<std::mutex> lk(g_timing_mu);
    if (begin) {
        TimedSpan s; s.key = key; s.e0 = timing_event(); s.e1 = timing_event(); s.closed = false;
        if (!s.e0 || !s.e1) return;
        (void)hipEventRecord(s.e0, st);
        g_spans.push_back(s);
    } else {
        for (size_t i = g_spans.size(); i-- > 0;)
            if (g_spans[i].key == key && !g_spans[i].closed) { (void)hipEventRecord(g_spans[i].e1, st); g_spans[i].closed = true; break; }
    }
}
extern "C" int pdp_kernel_timing(int enable)
{
    std::lock_guard<std::mutex> lk(g_timing_mu);
    for (auto &s : g_spans) { g_event_pool.push_back(s.e0); g_event_pool.push_back(s.e1); }
    g_spans.clear();
    g_pdp_timing_on = enable ? 1 : 0;
    return PDP_OK;
}
extern "C" int pdp_kernel_name(int key, char *buf, int len)
{
    PDP_REQUIRE(buf && len > 0 && key >= 0 && key < PDP_KN_COUNT, "bad argument");
    const char *n = g_pdp_kernel_name[key];
    snprintf(buf, (size_t)len, "%s", n ? n : "");
    return PDP_OK;
}
extern "C" int pdp_kernel_timing_read(float *ms_host, int32_t *launches_host)
{
    PDP_REQUIRE(ms_host && launches_host, "NULL argument");
    std::lock_guard<std::mutex> lk(g_timing_mu);
    for (int k = 0; k < PDP_TK_COUNT; ++k) { ms_host[k] = 0.0f; launches_host[k] = 0; }
    for (auto &s : g_spans) {
        if (s.closed && s.key >= 0 && s.key < PDP_TK_COUNT) {
            PDP_HIP_CHECK(hipEventSynchronize(s.e1));
            float ms = 0.0f;
            PDP_HIP_CHECK(hipEventElapsedTime(&ms, s.e0, s.e1));
            ms_host[s.key] += ms; launches_host[s.key] += 1;
        }
        g_event_pool.push_back(s.e0); g_event_pool.push_back(s.e1);
    }
    g_spans.clear();
    return PDP_OK;
}

// ---- kernels -----------------------------------------------------------------------------------------
__global__ void k_replicate(int E0, int V0, int F0, int B0, int R, const int32_t *gm, const int32_t *bvm,
                            const int32_t *bfm, const float *ef, int32_t *ogm, int32_t *ovi, int32_t *ofi,
                            float *osign, int8_t *osgn)
{
    const int64_t E = (int64_t)E0 * R, V = (int64_t)V0 * R, F = (int64_t)F0 * R;
    const int64_t total = E + V + F;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < E) {
            const int r = (int)(i / E0), e = (int)(i % E0);
            ogm[i] = gm[e] + r * V0;
            ogm[E + i] = gm[E0 + e] + r * F0;
            const float s = ef[e];
            osign[i] = s;
            osgn[i] = (int8_t)(s > 0.0f ? 1 : (s < 0.0f ? -1 : 0));
        } else if (i < E + V) {
            const int64_t j = i - E;
            const int r = (int)(j / V0), v = (int)(j % V0);
            ovi[j] = bvm[v] + r * B0;
        } else {
            const int64_t j = i - E - V;
            const int r = (int)(j / F0), c = (int)(j % F0);
            ofi[j] = bfm[c] + r * B0;
        }
    }
}

__global__ void k_validate(int E, int V, int F, int B, const int32_t *gm, const int32_t *vi, const int32_t *fi,
                           const float *sign, uint32_t *flags)
{
    const int64_t total = (int64_t)E + V + F;
    bool bad = false;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < E) {
            const int gv = gm[i], gf = gm[E + i];
            if (gv < 0 || gv >= V || gf < 0 || gf >= F) { bad = true; continue; }
            const int b = vi[gv];
            if (b != fi[gf]) bad = true;
            if (i > 0) { const int pv = gm[i - 1]; if (pv >= 0 && pv < V && vi[pv] > b) bad = true; }
            const float s = sign[i];
            if (!(s == 1.0f || s == -1.0f)) bad = true;
        } else if (i < (int64_t)E + V) {
            const int64_t j = i - E;
            const int b = vi[j];
            if (b < 0 || b >= B) bad = true;
            if (j > 0 && vi[j - 1] > b) bad = true;
        } else {
            const int64_t j = i - E - V;
            const int b = fi[j];
            if (b < 0 || b >= B) bad = true;
            if (j > 0 && fi[j - 1] > b) bad = true;
        }
    }
    if (bad) atomicOr(&flags[FL_LAYOUT_BAD], 1u);
}

__device__ __forceinline__ int lower_bound_i32(const int32_t *a, int n, int key)
{
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

// instance offsets: first variable / clause / edge with instance id >= b
__global__ void k_inst_offsets(int E, int V, int F, int B, const int32_t *gm, const int32_t *vi, const int32_t *fi,
                               int32_t *v0, int32_t *f0, int32_t *e0)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > B) return;
    v0[b] = lower_bound_i32(vi, V, b);
    f0[b] = lower_bound_i32(fi, F, b);
    int lo = 0, hi = E;   // edges are grouped by instance: key(e) = vi[gm[e]]
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (vi[gm[mid]] < b) lo = mid + 1; else hi = mid; }
    e0[b] = lo;
}

__global__ void k_local_ids(int E, const int32_t *gm, const int32_t *vi, const int32_t *v0, const int32_t *f0,
                            int32_t *e_var, int32_t *e_fn, int32_t *iota)
{
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const int gv = gm[e], gf = gm[E + e];
        const int b = vi[gv];
        e_var[e] = gv - v0[b];
        e_fn[e] = gf - f0[b];
        iota[e] = (int32_t)e;
    }
}

// rows = variables (or clauses): ptr[row + b] = lower_bound(sorted_keys, row) - e0[b]; edges[pos] = sorted_val - e0
__global__ void k_csr_finish(int E, int N, int B, const int32_t *sorted_keys, const int32_t *sorted_vals,
                             const int32_t *row_inst, const int32_t *row0, const int32_t *e0, int32_t *ptr,
                             int32_t *edges, uint32_t *identity_flag)
{
    const int64_t total = (int64_t)N + B + E;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < N) {
            const int b = row_inst[i];
            ptr[i + b] = lower_bound_i32(sorted_keys, E, (int)i) - e0[b];
        } else if (i < (int64_t)N + B) {
            const int b = (int)(i - N);
            ptr[row0[b + 1] + b] = e0[b + 1] - e0[b];       // terminal entry of instance b
        } else {
            const int64_t pos = i - N - B;
            const int b = row_inst[sorted_keys[pos]];
            edges[pos] = sorted_vals[pos] - e0[b];
            if (identity_flag && sorted_vals[pos] != (int32_t)pos) atomicOr(identity_flag, 1u);
        }
    }
}

// global-id CSR offsets: ptr[i] = first sorted position with key >= i, i in [0, N]
__global__ void k_global_ptr(int E, int N, const int32_t *sorted_keys, int32_t *ptr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= N) ptr[i] = lower_bound_i32(sorted_keys, E, i);
}

__global__ void k_max_dims(int B, const int32_t *v0, const int32_t *f0, const int32_t *e0, uint32_t *out /*[3]*/)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    atomicMax(&out[0], (uint32_t)(v0[b + 1] - v0[b]));
    atomicMax(&out[1], (uint32_t)(f0[b + 1] - f0[b]));
    atomicMax(&out[2], (uint32_t)(e0[b + 1] - e0[b]));
}

__global__ void k_fill(float *p, int64_t n, float v)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// SATProblem's fresh state (solver.py:49-54): active flags 1, solution 0.5, _is_sat 0.5, edge mask 1
__global__ void k_bind_fill(float *av, float *sol, int64_t V, float *af, int64_t F, float *is_sat, int64_t B, float *emask, int64_t E)
{
    const int64_t n = E > F ? (E > V ? E : V) : (F > V ? F : V);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < V) { av[i] = 1.0f; sol[i] = 0.5f; }
        if (i < F) af[i] = 1.0f;
        if (i < B) is_sat[i] = 0.5f;
        if (i < E) emask[i] = 1.0f;
    }
}

static inline int grid_for(int64_t n, int nt = 256) { int64_t g = (n + nt - 1) / nt; if (g < 1) g = 1; if (g > 8192) g = 8192; return (int)g; }

// ---- cached device allocator ---------------------------------------------------------------------------------------------------
#define PDP_POOL_LIMIT_BYTES ((size_t)16 << 30)     /* cached (free) bytes kept at most; the card has 288 GB */
namespace {
struct PoolBlock { void *p; size_t bytes; };
std::mutex g_pool_mu;
std::vector<PoolBlock> g_pool_free;
std::unordered_map<void *, size_t> g_pool_live;
size_t g_pool_cached = 0;
}

int pdp_dev_alloc(void **out, size_t bytes)
{
    *out = nullptr;
    const size_t need = ((bytes ? bytes : 1) + 4095) & ~(size_t)4095;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        size_t best = g_pool_free.size();
        for (size_t i = 0; i < g_pool_free.size(); ++i)             // smallest cached block of at least the size, at most 25 % larger
            if (g_pool_free[i].bytes >= need && g_pool_free[i].bytes <= need + need / 4 &&
                (best == g_pool_free.size() || g_pool_free[i].bytes < g_pool_free[best].bytes)) best = i;
        if (best != g_pool_free.size()) {
            const PoolBlock b = g_pool_free[best];
            g_pool_free[best] = g_pool_free.back(); g_pool_free.pop_back();
            g_pool_cached -= b.bytes;
            g_pool_live[b.p] = b.bytes;
            *out = b.p;
            return PDP_OK;
        }
    }
    void *ptr = nullptr;
    hipError_t e = hipMalloc(&ptr, need);
    if (e != hipSuccess) {
        // out of memory with blocks in the cache: give them back and retry once
        {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            for (const PoolBlock &b : g_pool_free) (void)hipFree(b.p);
            g_pool_free.clear(); g_pool_cached = 0;
        }
        (void)hipGetLastError();
        e = hipMalloc(&ptr, need);
    }
    if (e != hipSuccess) { pdp_set_error("hipMalloc(%zu bytes) failed: %s", need, hipGetErrorString(e)); return PDP_ERR_HIP; }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_live[ptr] = need;
    *out = ptr;
    return PDP_OK;
}

void pdp_dev_free(void *ptr)
{
    if (!ptr) return;
    std::vector<void *> evict;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_pool_live.find(ptr);
        if (it == g_pool_live.end()) { evict.push_back(ptr); }          // not ours (should not happen): plain free
        else {
            g_pool_free.push_back(PoolBlock{ptr, it->second});
            g_pool_cached += it->second;
            g_pool_live.erase(it);
            while (g_pool_cached > PDP_POOL_LIMIT_BYTES && !g_pool_free.empty()) {      // oldest first
                evict.push_back(g_pool_free.front().p); g_pool_cached -= g_pool_free.front().bytes;
                g_pool_free.erase(g_pool_free.begin());
            }
        }
    }
    for (void *q : evict) (void)hipFree(q);
}

template <typename T>
static int dmalloc(T **p, size_t n)
{
    return pdp_dev_alloc((void **)p, (n ? n : 1) * sizeof(T));
}

#define PDP_TRY(x) do { int _s = (x); if (_s != PDP_OK) return _s; } while (0)

extern "C" int pdp_problem_destroy(pdp_problem *p)
{
    if (!p) return PDP_OK;
    void *ptrs[] = {p->graph_map, p->var_inst, p->fn_inst, p->edge_sign, p->e_sgn, p->e_var, p->e_fn, p->inst_v0,
                    p->inst_f0, p->inst_e0, p->v_ptr, p->v_edges, p->f_ptr, p->f_edges, p->ws_e[0], p->ws_e[1],
                    p->ws_e[2], p->ws_e[3], p->ws_v[0], p->ws_v[1], p->ws_v[2], p->ws_v[3], p->ws_v[4], p->ws_v[5],
                    p->ws_f[0], p->ws_f[1], p->ws_b[0], p->ws_b[1], p->ws_b[2], p->ws_b[3], p->ws_bi[0], p->ws_bi[1],
                    p->ws_vi[0], p->ws_vi[1], p->ws_vi[2], p->ws_fu[0], p->ws_fu[1], p->flags, p->cub_tmp};
    for (void *q : ptrs) if (q) pdp_dev_free(q);
    if (p->flags_host) (void)hipHostFree(p->flags_host);
    if (p->solve_blob) pdp_dev_free(p->solve_blob);
    if (p->solve_host) (void)hipHostFree(p->solve_host);
    if (p->exchange_host) (void)hipHostFree(p->exchange_host);
    if (p->solve_extra_v) pdp_dev_free(p->solve_extra_v);
    if (p->solve_rec) pdp_dev_free(p->solve_rec);
    void *res[] = {p->res_stat_off, p->res_stat, p->res_dyn[0], p->res_dyn[1], p->res_prev_slots, p->res_ctl, p->res_fit_list, p->res_big_list, p->res_is_big, p->res_big_snap};
    for (void *q : res) if (q) pdp_dev_free(q);
    if (p->simp_topo) pdp_dev_free(p->simp_topo);
    if (p->team_ws) pdp_dev_free(p->team_ws);
    if (p->ws_fit_list) pdp_dev_free(p->ws_fit_list);
    if (p->ws_big_list) pdp_dev_free(p->ws_big_list);
    if (p->ws_big_off) pdp_dev_free(p->ws_big_off);
    if (p->ws_side_stream) { (void)hipStreamDestroy(p->ws_side_stream); for (int i = 0; i < 2; ++i) (void)hipEventDestroy(p->ws_side_ev[i]); }
    if (p->res_side_stream) { (void)hipStreamDestroy(p->res_side_stream); for (int i = 0; i < 2; ++i) (void)hipEventDestroy(p->res_side_ev[i]); }
    for (int i = 0; i < p->res_events_n; ++i) (void)hipEventDestroy(p->res_events[i]);
    free(p->res_events);
    for (int i = 0; i < 4; ++i) if (p->nws[i]) pdp_dev_free(p->nws[i]);
    if (p->nv_ptr) pdp_dev_free(p->nv_ptr);
    if (p->nv_edges) pdp_dev_free(p->nv_edges);
    if (p->nf_ptr) pdp_dev_free(p->nf_ptr);
    if (p->nf_edges) pdp_dev_free(p->nf_edges);
    delete p;
    return PDP_OK;
}

static int build_problem(pdp_problem *p, const int32_t *graph_map, const int32_t *bvm, const int32_t *bfm,
                         const float *edge_feature, hipStream_t st)
{
    const int E = p->E, V = p->V, F = p->F, B = p->B;
    PDP_TRY(dmalloc(&p->graph_map, (size_t)2 * E));
    PDP_TRY(dmalloc(&p->var_inst, (size_t)V));
    PDP_TRY(dmalloc(&p->fn_inst, (size_t)F));
    PDP_TRY(dmalloc(&p->edge_sign, (size_t)E));
    PDP_TRY(dmalloc(&p->e_sgn, (size_t)E));
    PDP_TRY(dmalloc(&p->e_var, (size_t)E));
    PDP_TRY(dmalloc(&p->e_fn, (size_t)E));
    PDP_TRY(dmalloc(&p->inst_v0, (size_t)B + 1));
    PDP_TRY(dmalloc(&p->inst_f0, (size_t)B + 1));
    PDP_TRY(dmalloc(&p->inst_e0, (size_t)B + 1));
    PDP_TRY(dmalloc(&p->v_ptr, (size_t)V + B));
    PDP_TRY(dmalloc(&p->v_edges, (size_t)E));
    PDP_TRY(dmalloc(&p->f_ptr, (size_t)F + B));
    PDP_TRY(dmalloc(&p->f_edges, (size_t)E));
    for (int i = 0; i < 4; ++i) PDP_TRY(dmalloc(&p->ws_e[i], (size_t)E));
    for (int i = 0; i < 6; ++i) PDP_TRY(dmalloc(&p->ws_v[i], (size_t)V));
    for (int i = 0; i < 2; ++i) PDP_TRY(dmalloc(&p->ws_f[i], (size_t)F));
    for (int i = 0; i < 4; ++i) PDP_TRY(dmalloc(&p->ws_b[i], (size_t)B));
    for (int i = 0; i < 2; ++i) PDP_TRY(dmalloc(&p->ws_bi[i], (size_t)B));
    for (int i = 0; i < 3; ++i) PDP_TRY(dmalloc(&p->ws_vi[i], (size_t)V));
    for (int i = 0; i < 2; ++i) PDP_TRY(dmalloc(&p->ws_fu[i], (size_t)F));
    PDP_TRY(dmalloc(&p->flags, (size_t)FL_COUNT));
    PDP_HIP_CHECK(hipHostMalloc((void **)&p->flags_host, FL_COUNT * sizeof(uint32_t)));
    PDP_HIP_CHECK(hipMemsetAsync(p->flags, 0, FL_COUNT * sizeof(uint32_t), st));

    hipLaunchKernelGGL(k_replicate, dim3(grid_for((int64_t)E + V + F)), dim3(256), 0, st, p->E0, p->V0, p->F0, p->B0,
                       p->R, graph_map, bvm, bfm, edge_feature, p->graph_map, p->var_inst, p->fn_inst, p->edge_sign, p->e_sgn);
    PDP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_validate, dim3(grid_for((int64_t)E + V + F)), dim3(256), 0, st, E, V, F, B, p->graph_map,
                       p->var_inst, p->fn_inst, p->edge_sign, p->flags);
    PDP_LAUNCH_CHECK();
    PDP_HIP_CHECK(hipMemcpyAsync(p->flags_host, p->flags, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    PDP_HIP_CHECK(hipStreamSynchronize(st));
    if (p->flags_host[FL_LAYOUT_BAD]) {
        pdp_set_error("batch is not in the loader's instance-contiguous layout (sorted batch maps, edges grouped by "
                      "instance, signs in {-1,+1}, ids in range)");
        return PDP_ERR_LAYOUT;
    }
    hipLaunchKernelGGL(k_inst_offsets, dim3((B + 1 + 255) / 256), dim3(256), 0, st, E, V, F, B, p->graph_map, p->var_inst,
                       p->fn_inst, p->inst_v0, p->inst_f0, p->inst_e0);
    PDP_LAUNCH_CHECK();

    // stable sorts of (row id, edge id) give both CSR structures with ascending edge ids per row
    int32_t *iota = (int32_t *)p->ws_vi[0];   // V ints are not enough for E: use dedicated temporaries
    int32_t *keys_out = nullptr, *vals_out = nullptr, *iota_e = nullptr;
    PDP_TRY(dmalloc(&keys_out, (size_t)E));
    PDP_TRY(dmalloc(&vals_out, (size_t)E));
    PDP_TRY(dmalloc(&iota_e, (size_t)E));
    (void)iota;
    hipLaunchKernelGGL(k_local_ids, dim3(grid_for(E)), dim3(256), 0, st, E, p->graph_map, p->var_inst, p->inst_v0,
                       p->inst_f0, p->e_var, p->e_fn, iota_e);
    PDP_LAUNCH_CHECK();
    size_t tmp_bytes = 0;
    int bits_v = 1; while ((1ll << bits_v) < (long long)V) ++bits_v;
    int bits_f = 1; while ((1ll << bits_f) < (long long)F) ++bits_f;
    const int bits = bits_v > bits_f ? bits_v : bits_f;
    PDP_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, p->graph_map, keys_out, iota_e, vals_out, E, 0, bits, st));
    PDP_TRY(dmalloc((char **)&p->cub_tmp, tmp_bytes));
    p->cub_tmp_bytes = tmp_bytes;
    // by variable
    PDP_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(p->cub_tmp, tmp_bytes, p->graph_map, keys_out, iota_e, vals_out, E, 0, bits, st));
    hipLaunchKernelGGL(k_csr_finish, dim3(grid_for((int64_t)V + B + E)), dim3(256), 0, st, E, V, B, keys_out, vals_out,
                       p->var_inst, p->inst_v0, p->inst_e0, p->v_ptr, p->v_edges, (uint32_t *)nullptr);
    PDP_LAUNCH_CHECK();
    PDP_TRY(dmalloc(&p->nv_ptr, (size_t)V + 1)); PDP_TRY(dmalloc(&p->nv_edges, (size_t)E));
    hipLaunchKernelGGL(k_global_ptr, dim3((V + 1 + 255) / 256), dim3(256), 0, st, E, V, keys_out, p->nv_ptr);
    PDP_HIP_CHECK(hipMemcpyAsync(p->nv_edges, vals_out, sizeof(int32_t) * (size_t)E, hipMemcpyDeviceToDevice, st));
    // by clause
    PDP_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(p->cub_tmp, tmp_bytes, p->graph_map + E, keys_out, iota_e, vals_out, E, 0, bits, st));
    hipLaunchKernelGGL(k_csr_finish, dim3(grid_for((int64_t)F + B + E)), dim3(256), 0, st, E, F, B, keys_out, vals_out,
                       p->fn_inst, p->inst_f0, p->inst_e0, p->f_ptr, p->f_edges, p->flags + FL_GMIN0);
    PDP_LAUNCH_CHECK();
    PDP_TRY(dmalloc(&p->nf_ptr, (size_t)F + 1)); PDP_TRY(dmalloc(&p->nf_edges, (size_t)E));
    hipLaunchKernelGGL(k_global_ptr, dim3((F + 1 + 255) / 256), dim3(256), 0, st, E, F, keys_out, p->nf_ptr);
    PDP_HIP_CHECK(hipMemcpyAsync(p->nf_edges, vals_out, sizeof(int32_t) * (size_t)E, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_max_dims, dim3((B + 255) / 256), dim3(256), 0, st, B, p->inst_v0, p->inst_f0, p->inst_e0,
                       p->flags + FL_GMIN1);
    PDP_LAUNCH_CHECK();
    PDP_HIP_CHECK(hipMemcpyAsync(p->flags_host, p->flags, FL_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    PDP_HIP_CHECK(hipStreamSynchronize(st));
    p->fn_edges_identity = p->flags_host[FL_GMIN0] ? 0 : 1;
    p->max_n = (int)p->flags_host[FL_GMIN1];
    p->max_m = (int)p->flags_host[FL_GMIN1 + 1];
    p->max_e = (int)p->flags_host[FL_GMIN1 + 2];
    PDP_HIP_CHECK(hipMemsetAsync(p->flags, 0, FL_COUNT * sizeof(uint32_t), st));
    pdp_dev_free(keys_out); pdp_dev_free(vals_out); pdp_dev_free(iota_e);
    return PDP_OK;
}

extern "C" int pdp_problem_create(pdp_problem **out, int E, int V, int F, int B, int replication,
                                  const int32_t *graph_map, const int32_t *batch_variable_map,
                                  const int32_t *batch_function_map, const float *edge_feature, void *stream)
{
    PDP_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    PDP_REQUIRE(E > 0 && V > 0 && F > 0 && B > 0, "empty batch");
    PDP_REQUIRE(graph_map && batch_variable_map && batch_function_map && edge_feature, "NULL input array");
    const int R = replication < 1 ? 1 : replication;
    PDP_REQUIRE((int64_t)E * R < (1ll << 31) && (int64_t)F * R < (1ll << 31), "batch too large for int32 ids");
    int ndev = 0;
    PDP_HIP_CHECK(hipGetDeviceCount(&ndev));
    pdp_problem *p = new pdp_problem();
    memset(p, 0, sizeof(*p));
    p->E0 = E; p->V0 = V; p->F0 = F; p->B0 = B; p->R = R;
    p->E = E * R; p->V = V * R; p->F = F * R; p->B = B * R;
    const int s = build_problem(p, graph_map, batch_variable_map, batch_function_map, edge_feature, (hipStream_t)stream);
    if (s != PDP_OK) { pdp_problem_destroy(p); return s; }
    *out = p;
    return PDP_OK;
}

extern "C" int pdp_problem_set_rng_base(pdp_problem *p, uint32_t first_variable, uint32_t first_instance)
{
    PDP_REQUIRE(p, "NULL problem");
    PDP_REQUIRE(p->R == 1 || (first_variable == 0u && first_instance == 0u), "a part of a forward cannot be replicated (replica r of variable v has index v + r * V)");
    p->rng_var_base = first_variable; p->rng_inst_base = first_instance;
    return PDP_OK;
}

extern "C" int pdp_problem_set_exchange(pdp_problem *p, int (*fn)(void *, uint32_t *, int, uint32_t *, int, uint32_t *, int), void *user)
{
    PDP_REQUIRE(p, "NULL problem");
    PDP_REQUIRE(!fn || p->R == 1, "a coupled forward over several processes cannot be replicated");
    p->exchange = fn; p->exchange_user = user;
    return PDP_OK;
}
// the staging block of the exchange: mins | maxs | ors in one pinned allocation
int pdp_exchange_call(pdp_problem *p, const uint32_t *mins, int n_mins, const uint32_t *maxs, int n_maxs, const uint32_t *ors, int n_ors, uint32_t **out)
{
    const size_t words = (size_t)n_mins + n_maxs + n_ors;
    if (p->exchange_host_words < words) {
        if (p->exchange_host) (void)hipHostFree(p->exchange_host);
        p->exchange_host = nullptr; p->exchange_host_words = 0;
        PDP_HIP_CHECK(hipHostMalloc((void **)&p->exchange_host, (words + 64) * 4));
        p->exchange_host_words = words + 64;
    }
    uint32_t *h = p->exchange_host;
    for (int i = 0; i < n_mins; ++i) h[i] = mins[i];
    for (int i = 0; i < n_maxs; ++i) h[n_mins + i] = maxs[i];
    for (int i = 0; i < n_ors; ++i) h[n_mins + n_maxs + i] = ors[i];
    const int rc = p->exchange(p->exchange_user, h, n_mins, h + n_mins, n_maxs, h + n_mins + n_maxs, n_ors);
    if (rc != 0) { pdp_set_error("the exchange callback of a coupled multi-process forward failed"); return PDP_ERR_INVALID; }
    *out = h;
    return PDP_OK;
}

extern "C" int pdp_problem_dims(const pdp_problem *p, int32_t *d)
{
    PDP_REQUIRE(p && d, "NULL argument");
    d[0] = p->E; d[1] = p->V; d[2] = p->F; d[3] = p->B; d[4] = p->R; d[5] = p->max_n; d[6] = p->max_m; d[7] = p->max_e;
    return PDP_OK;
}

extern "C" int pdp_problem_export_graph(const pdp_problem *p, int32_t *graph_map, int32_t *bvm, int32_t *bfm,
                                        float *edge_feature, void *stream)
{
    PDP_REQUIRE(p, "NULL problem");
    hipStream_t st = (hipStream_t)stream;
    if (graph_map) PDP_HIP_CHECK(hipMemcpyAsync(graph_map, p->graph_map, sizeof(int32_t) * 2 * (size_t)p->E, hipMemcpyDeviceToDevice, st));
    if (bvm) PDP_HIP_CHECK(hipMemcpyAsync(bvm, p->var_inst, sizeof(int32_t) * (size_t)p->V, hipMemcpyDeviceToDevice, st));
    if (bfm) PDP_HIP_CHECK(hipMemcpyAsync(bfm, p->fn_inst, sizeof(int32_t) * (size_t)p->F, hipMemcpyDeviceToDevice, st));
    if (edge_feature) PDP_HIP_CHECK(hipMemcpyAsync(edge_feature, p->edge_sign, sizeof(float) * (size_t)p->E, hipMemcpyDeviceToDevice, st));
    return PDP_OK;
}

extern "C" int pdp_problem_bind_state(pdp_problem *p, float *av, float *af, float *sol, float *is_sat, float *emask, void *stream)
{
    PDP_REQUIRE(p && av && af && sol && is_sat && emask, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    p->av = av; p->af = af; p->sol = sol; p->is_sat = is_sat; p->emask = emask; p->has_edge_mask = 0;
    // one launch for the five arrays (five fills were five launches and their gaps: ~45 us of a 9 ms headline step)
    hipLaunchKernelGGL(k_bind_fill, dim3(grid_for(p->E > p->F ? p->E : p->F)), dim3(256), 0, st, av, sol, (int64_t)p->V, af, (int64_t)p->F, is_sat, (int64_t)p->B,
                       emask, (int64_t)p->E);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// Team plan of a launch over `count` instances with `threads` per workgroup (see pdp_solve.hip::launch_hbm for the rules): out->size == 1
// means one workgroup per instance.  Allocates and clears the team workspace on `st`.
int pdp_team_plan(pdp_problem *p, int count, bool wide, int threads, TeamLaunch *out, hipStream_t st)
{
    int cap = PDP_TEAM_MAX;
    if (const char *env = getenv("PDP_SOLVE_TEAM")) { const int v = atoi(env); if (v >= 1 && v <= PDP_TEAM_MAX) cap = v; }
    size_t wide_edges = 300000;
    if (const char *env = getenv("PDP_SOLVE_TEAM_WIDE_EDGES")) wide_edges = (size_t)atoll(env);
    // workgroups that are certainly resident together: one per CU of the device (a team's workgroups wait for each other)
    const int cus = pdp_device_cus();
    const int per_xcd_cus = cus / 8 > 0 ? cus / 8 : 1;
    const bool go_wide = wide && wide_edges > 0 && (size_t)p->max_e >= wide_edges && count * 2 <= cus;
    int size = 1;
    const int per_xcd = ((count + 7) & ~7) / 8;          // teams that share an XCD
    if (go_wide) { while (size * 2 <= cap && (size_t)count * size * 2 <= (size_t)cus && (size_t)p->max_e >= (size_t)size * 2 * threads * 2) size *= 2; }
    else { while (size * 2 <= cap && per_xcd * size * 2 <= per_xcd_cus && (size_t)p->max_e >= (size_t)size * 2 * threads * 2) size *= 2; }
    out->size = size; out->count = count; out->no_xcd = getenv("PDP_SOLVE_TEAM_AGENT_FENCES") ? 1 : 0; out->ws = nullptr;
    out->spin_limit = pdp_spin_limit();
    // slot-minor numbering; one-XCD teams pad the slot count to the XCD count so that a team's workgroups share an XCD
    out->slots = go_wide ? count : ((count + 7) & ~7);
    if (size > 1) {
        if (!p->team_ws) { int st_ = pdp_dev_alloc((void **)&p->team_ws, sizeof(uint32_t) * 256 * PDP_TEAM_WORDS); if (st_ != PDP_OK) return st_; }
        PDP_HIP_CHECK(hipMemsetAsync(p->team_ws, 0, sizeof(uint32_t) * (size_t)count * PDP_TEAM_WORDS, st));
        out->ws = p->team_ws;
    }
    return PDP_OK;
}

uint32_t pdp_spin_limit()
{
    if (const char *env = getenv("PDP_TEAM_SPIN_LIMIT")) { const long long v = atoll(env); if (v >= 1024 && v <= 0xffffffffll) return (uint32_t)v; }
    return PDP_SPIN_LIMIT_DEFAULT;
}

int pdp_edge_rows(const pdp_problem *p)
{
    if (p->max_e < 16384) return 1;
    // ~8 edges per thread for the largest instance, within a total of ~64k workgroups
    long rows = ((long)p->max_e + 8L * PDP_NT - 1) / (8L * PDP_NT);
    const long cap = 65535L / (p->B > 0 ? p->B : 1);
    if (rows > cap) rows = cap;
    if (rows > 1024) rows = 1024;
    return rows < 1 ? 1 : (int)rows;
}

int pdp_device_cus()
{
    // per device id (one process may drive several devices; a count cached for the first one would size co-resident launches -- teams,
    // the lock-step kernel -- for the wrong chip and their spin barriers would never complete)
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 64;
    if (!cus[dev]) { int v = 0; cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 64; }
    return cus[dev];
}

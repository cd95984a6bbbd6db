// pdp_coo.hip -- the reference's L0 primitives in their ORIGINAL call shape: util.sparse_max / sparse_argmax / sparse_smooth_max and
// torch.mm(mask, X) with the mask handed over as the index / value arrays of a sparse COO matrix (reference: src/pdp/nn/util.py:257-286,
// :60-69).  A plug-in written against the reference passes such masks (sat_problem._batch_mask_tuple[0], masks it built itself with
// SatLossEvaluator.compute_masks ...) instead of a problem handle; the masks SATProblem built are mapped back to the resident layout on
// the host side (pdp/nn/util.py), everything else lands here.  No pdp_problem is involved: the entry points work on any incidence
// structure.  Determinism: the max / arg-max forms reduce 64-bit keys with integer atomics (order independent); the products run on
// row-sorted entries and add them in ascending entry order, like a coalesced torch sparse product on the host.
#include "pdp_common.hpp"

#define ST(s) ((hipStream_t)(s))

static inline int coo_grid(int64_t n, int nt = 256) { int64_t g = (n + nt - 1) / nt; if (g < 1) g = 1; if (g > 8192) g = 8192; return (int)g; }

// order-preserving map of a float onto uint32 (NaN above +inf: torch.max / torch.argmax treat NaN as the largest value)
__device__ __forceinline__ uint32_t coo_key(float v)
{
    if (v != v) return 0xffffffffu;
    const uint32_t u = pdp_f2bits(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float coo_unkey(uint32_t k)
{
    if (k == 0xffffffffu) return PDP_NAN;
    return pdp_bits2f((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// scratch[0] = ordered key of the minimum of x, scratch[1] = a NaN was seen; scratch[2 ..] = one key per group
__global__ void k_coo_init(uint64_t *scratch, int64_t groups)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i == 0) { scratch[0] = 0xffffffffull; scratch[1] = 0ull; }
    for (int64_t g = i; g < groups; g += (int64_t)gridDim.x * blockDim.x) scratch[2 + g] = 0ull;
}

__global__ void __launch_bounds__(256) k_coo_min(const float *x, int64_t n, uint64_t *scratch)
{
    __shared__ uint32_t smin; __shared__ uint32_t snan;
    if (threadIdx.x == 0) { smin = 0xffffffffu; snan = 0u; }
    __syncthreads();
    uint32_t m = 0xffffffffu; bool nn = false;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        if (v != v) nn = true; else { const uint32_t k = coo_key(v); m = k < m ? k : m; }
    }
    atomicMin(&smin, m);
    if (nn) atomicOr(&snan, 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin((unsigned long long *)&scratch[0], (unsigned long long)smin);
        if (snan) atomicOr((unsigned long long *)&scratch[1], 1ull);
    }
}

// dense_mat[rows[i], cols[i]] = x[i] - x.min() + 1 (util.py:260-263, :270-273): entry i of the mask's index list carries x[i].
// key = (ordered value << 32) | (2^32 - 1 - row): one atomicMax per entry leaves the largest value with the SMALLEST row on ties,
// which is torch.argmax's first-occurrence rule along dim 0
__global__ void __launch_bounds__(256) k_coo_scatter(const int64_t *rows, const int64_t *cols, int64_t nnz, const float *x, int64_t n_cols,
                                                    uint64_t *scratch)
{
    const float gmin = scratch[1] ? PDP_NAN : coo_unkey((uint32_t)scratch[0]);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = cols[i];
        if (c < 0 || c >= n_cols) continue;
        const float t = (x[i] - gmin) + 1.0f;
        const uint64_t key = ((uint64_t)coo_key(t) << 32) | (uint64_t)(0xffffffffu - (uint32_t)rows[i]);
        atomicMax((unsigned long long *)&scratch[2 + c], (unsigned long long)key);
    }
}

__global__ void k_coo_finish_max(const uint64_t *scratch, int64_t n_cols, float *out)
{
    const float gmin = scratch[1] ? PDP_NAN : coo_unkey((uint32_t)scratch[0]);
    for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < n_cols; g += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = scratch[2 + g];
        // a column of the dense matrix also holds the zeros of the rows outside the group; an entry is x - min + 1 >= 1 (or NaN, which
        // wins either way), so the zeros only show in an empty column
        const float t = key ? coo_unkey((uint32_t)(key >> 32)) : 0.0f;
        out[g] = (t + gmin) - 1.0f;
    }
}

__global__ void k_coo_finish_argmax(const uint64_t *scratch, int64_t n_cols, int64_t *out)
{
    for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < n_cols; g += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = scratch[2 + g];
        // an empty column is all zeros: arg-max 0.  A column whose best entry is below zero (x - min + 1 < 0 cannot happen for finite x)
        out[g] = key ? (int64_t)(0xffffffffu - (uint32_t)(key & 0xffffffffull)) : 0;
    }
}

static int coo_reduce(const int64_t *rows, const int64_t *cols, int64_t nnz, const float *x, int64_t n_rows, int64_t n_cols, uint64_t *scratch,
                      hipStream_t st)
{
    PDP_REQUIRE(nnz >= 0 && n_cols >= 0 && n_rows >= 0 && n_rows < (int64_t)0xffffffffll, "bad sizes");
    PDP_REQUIRE(scratch && (nnz == 0 || (rows && cols && x)), "NULL argument");
    hipLaunchKernelGGL(k_coo_init, dim3(coo_grid(n_cols)), dim3(256), 0, st, scratch, n_cols);
    if (nnz > 0) {
        hipLaunchKernelGGL(k_coo_min, dim3(coo_grid(nnz)), dim3(256), 0, st, x, nnz, scratch);
        hipLaunchKernelGGL(k_coo_scatter, dim3(coo_grid(nnz)), dim3(256), 0, st, rows, cols, nnz, x, n_cols, scratch);
    }
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

extern "C" int pdp_coo_max(const int64_t *rows, const int64_t *cols, int64_t nnz, const float *x, int64_t n_rows, int64_t n_cols, uint64_t *scratch,
                           float *out, void *stream)
{
    PDP_REQUIRE(out || n_cols == 0, "NULL argument");
    const int s = coo_reduce(rows, cols, nnz, x, n_rows, n_cols, scratch, ST(stream));
    if (s != PDP_OK) return s;
    hipLaunchKernelGGL(k_coo_finish_max, dim3(coo_grid(n_cols)), dim3(256), 0, ST(stream), scratch, n_cols, out);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

extern "C" int pdp_coo_argmax(const int64_t *rows, const int64_t *cols, int64_t nnz, const float *x, int64_t n_rows, int64_t n_cols, uint64_t *scratch,
                              int64_t *out, void *stream)
{
    PDP_REQUIRE(out || n_cols == 0, "NULL argument");
    const int s = coo_reduce(rows, cols, nnz, x, n_rows, n_cols, scratch, ST(stream));
    if (s != PDP_OK) return s;
    hipLaunchKernelGGL(k_coo_finish_argmax, dim3(coo_grid(n_cols)), dim3(256), 0, ST(stream), scratch, n_cols, out);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// ---- products on row-sorted entries ------------------------------------------------------------------------------------------------
// row_ptr [n_rows + 1] of entries sorted by row: row_ptr[r] = first entry whose row is >= r
__global__ void __launch_bounds__(256) k_coo_row_ptr(const int64_t *rows, int64_t nnz, int64_t n_rows, int64_t *row_ptr)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i <= nnz; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t lo = (i == 0) ? 0 : rows[i - 1] + 1;
        const int64_t hi = (i == nnz) ? n_rows : rows[i];
        for (int64_t r = lo; r <= hi && r <= n_rows; ++r) row_ptr[r] = i;
    }
}

extern "C" int pdp_coo_row_ptr(const int64_t *sorted_rows, int64_t nnz, int64_t n_rows, int64_t *row_ptr, void *stream)
{
    PDP_REQUIRE(row_ptr && nnz >= 0 && n_rows >= 0 && (nnz == 0 || sorted_rows), "bad argument");
    hipLaunchKernelGGL(k_coo_row_ptr, dim3(coo_grid(nnz + 1)), dim3(256), 0, ST(stream), sorted_rows, nnz, n_rows, row_ptr);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// out [n_rows, d] = mask X: one thread per (row, column); a row's entries are added in ascending entry order.  `sub` (optional, [n_rows, d])
// is subtracted afterwards: the exclude-self form mm(mask_transpose, aggregated) - state of util.py:63-69 in one pass
__global__ void __launch_bounds__(256) k_csr_matmul(const int64_t *row_ptr, const int64_t *cols, const float *vals, int64_t n_rows, const float *X,
                                                   int d, int64_t ldx, const float *sub, float *out)
{
    const int64_t total = n_rows * (int64_t)d;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / d; const int j = (int)(t - r * d);
        float acc = 0.0f;
        for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k) {
            const float xv = X[cols[k] * ldx + j];
            acc = acc + (vals ? vals[k] * xv : xv);
        }
        if (sub) acc = acc - sub[t];
        out[t] = acc;
    }
}

extern "C" int pdp_csr_matmul(const int64_t *row_ptr, const int64_t *cols, const float *vals, int64_t n_rows, const float *X, int d, int64_t ldx,
                              const float *sub, float *out, void *stream)
{
    PDP_REQUIRE(row_ptr && n_rows >= 0 && d >= 0 && ldx >= d, "bad argument");
    if (n_rows == 0 || d == 0) return PDP_OK;
    PDP_REQUIRE(out && X && cols, "NULL argument");
    hipLaunchKernelGGL(k_csr_matmul, dim3(coo_grid(n_rows * (int64_t)d)), dim3(256), 0, ST(stream), row_ptr, cols, vals, n_rows, X, d, ldx, sub, out);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

// util.sparse_smooth_max (util.py:282-286) on row-sorted entries: out[r] = sum(m x coeff) / max(sum(m coeff), 1), coeff = exp(min(alpha x, 30))
__global__ void __launch_bounds__(256) k_csr_smooth_max(const int64_t *row_ptr, const int64_t *cols, const float *vals, int64_t n_rows, const float *x,
                                                       float alpha, float *out)
{
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x) {
        float num = 0.0f, den = 0.0f;
        for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k) {
            const float xe = x[cols[k]];
            const float coeff = pdp_safe_exp(alpha * xe);
            const float m = vals ? vals[k] : 1.0f;
            num = num + m * (xe * coeff);
            den = den + m * coeff;
        }
        out[r] = num / pdp_max(den, 1.0f);
    }
}

extern "C" int pdp_csr_smooth_max(const int64_t *row_ptr, const int64_t *cols, const float *vals, int64_t n_rows, const float *x, float alpha,
                                  float *out, void *stream)
{
    PDP_REQUIRE(row_ptr && n_rows >= 0, "bad argument");
    if (n_rows == 0) return PDP_OK;
    PDP_REQUIRE(out && x && cols, "NULL argument");
    hipLaunchKernelGGL(k_csr_smooth_max, dim3(coo_grid(n_rows)), dim3(256), 0, ST(stream), row_ptr, cols, vals, n_rows, x, alpha, out);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

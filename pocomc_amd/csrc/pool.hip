// Device side of the SMC bookkeeping around the MCMC step: what pocomc/sampler.py does with numpy on the persistent
// particle pool between two mutations -- importance weights (sampler.py:779-781), trimming (tools.py:38-41), the
// geometry of theta (geometry.py:31-59, student.py:43-48) and the bootstrap of the evidence (sampler.py:905-911) -- on a
// pool that stays in HBM.  Reductions are ordered (block partials summed by one block in index order): a run is
// reproducible bit for bit.
#include <hip/hip_runtime.h>
#include <string.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>
#include "philox.h"
#include "pmc_internal.h"

#pragma clang fp contract(off)

// ---------------------------------------------------------------------------------------------------------------
// weights = exp(logw - max) / sum   (sampler.py:779-781; max / sum from pmc_logw_stats)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void weights_kernel(const double* __restrict__ logw, int64_t P, const double* __restrict__ stats,
                                                      double* __restrict__ w) {
    const double mx = stats[0], sm = stats[1];
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < P; e += (int64_t)gridDim.x * 256) w[e] = exp(logw[e] - mx) / sm;
}

extern "C" int pmc_weights_from_logw(const double* logw, int64_t P, const double* stats, double* w, void* stream) {
    if (!logw || !stats || !w || P < 1) return pmc_fail("pmc_weights_from_logw: bad argument");
    int64_t grid = (P + 255) / 256; if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(weights_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, logw, P, stats, w);
    return pmc_check_launch("weights_kernel");
}

// ---------------------------------------------------------------------------------------------------------------
// trim_weights, tools.py:38-41: keep w >= threshold (order preserved), renormalise
// ---------------------------------------------------------------------------------------------------------------
struct KeepAbove {
    const double* w; const double* thr;
    __device__ __forceinline__ bool operator()(const int64_t& i) const { return w[i] >= thr[0]; }
};

__global__ __launch_bounds__(256) void take_kernel(const double* __restrict__ w, const int64_t* __restrict__ idx,
                                                   const int64_t* __restrict__ count, double* __restrict__ out) {
    const int64_t n = *count;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) out[e] = w[idx[e]];
}

// ordered sum of a[0 .. *count) by one workgroup (P <= a few 1e5: a few microseconds)
__global__ __launch_bounds__(1024) void ordered_sum_kernel(const double* __restrict__ a, const int64_t* __restrict__ count,
                                                           double* __restrict__ total) {
    __shared__ double part[1024];
    const int64_t n = *count;
    const int64_t per = (n + 1023) / 1024;
    double s = 0.0;
    const int64_t lo = (int64_t)threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (int64_t e = lo; e < hi; ++e) s += a[e];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = part[0];
}

__global__ __launch_bounds__(256) void divide_kernel(double* __restrict__ a, const int64_t* __restrict__ count,
                                                     const double* __restrict__ total) {
    const int64_t n = *count;
    const double t = *total;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) a[e] /= t;
}

__global__ __launch_bounds__(1024) void ordered_sum_n_kernel(const double* __restrict__ a, int64_t n, double* __restrict__ total) {
    __shared__ double part[1024];
    const int64_t per = (n + 1023) / 1024;
    double s = 0.0;
    const int64_t lo = (int64_t)threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (int64_t e = lo; e < hi; ++e) s += a[e];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = part[0];
}

// out f64 [1] (device) <- sum a[0..n), added in a fixed order
extern "C" int pmc_sum_f64(const double* a, int64_t n, double* out, void* stream) {
    if (!a || !out || n < 1) return pmc_fail("pmc_sum_f64: bad argument");
    hipLaunchKernelGGL(ordered_sum_n_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, n, out);
    return pmc_check_launch("ordered_sum_n_kernel");
}

extern "C" int64_t pmc_trim_select_workspace_bytes(int64_t P) {
    size_t tmp = 0;
    rocprim::counting_iterator<int64_t> it(0);
    KeepAbove op{nullptr, nullptr};
    (void)rocprim::select(nullptr, tmp, it, (int64_t*)nullptr, (int64_t*)nullptr, (size_t)P, op);
    return (int64_t)(tmp + 256);
}

// idx_out i64 [<= P], w_out f64 [<= P], count i64 [1] (device), threshold f64 [1] (device: result[0] of pmc_trim_threshold)
extern "C" int pmc_trim_select(const double* w, int64_t P, const double* threshold, int64_t* idx_out, double* w_out,
                               int64_t* count, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!w || !threshold || !idx_out || !w_out || !count || !workspace || P < 1) return pmc_fail("pmc_trim_select: bad argument");
    hipStream_t st = (hipStream_t)stream;
    size_t need = 0;
    rocprim::counting_iterator<int64_t> it(0);
    KeepAbove op{w, threshold};
    (void)rocprim::select(nullptr, need, it, idx_out, count, (size_t)P, op);
    if ((int64_t)need + 16 > workspace_bytes) return pmc_fail("pmc_trim_select: workspace too small");
    double* total = (double*)workspace;
    void* tmp = (void*)((char*)workspace + 16);
    if (rocprim::select(tmp, need, it, idx_out, count, (size_t)P, op, st) != hipSuccess)
        return pmc_fail("pmc_trim_select: select failed");
    int64_t grid = (P + 255) / 256; if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(take_kernel, dim3((unsigned)grid), dim3(256), 0, st, w, (const int64_t*)idx_out, (const int64_t*)count, w_out);
    hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(1024), 0, st, (const double*)w_out, (const int64_t*)count, total);
    hipLaunchKernelGGL(divide_kernel, dim3((unsigned)grid), dim3(256), 0, st, w_out, (const int64_t*)count, (const double*)total);
    return pmc_check_launch("pmc_trim_select");
}

// ---------------------------------------------------------------------------------------------------------------
// first and second moments of rows x[idx[r]] (idx optional) with weights w (optional):
//   out[0] = V1 = sum w, out[1] = V2 = sum w^2, mean[j] = sum w x_j / V1, S[i][j] = sum w (x_i - mean_i)(x_j - mean_j)
// -- everything np.average / np.cov(aweights) / np.var need (geometry.py:44-49, student.py:47).  Input rows float64 or
// float32 (theta is the float32 output of the flow, tools.py:336-340); accumulation in float64.
// ---------------------------------------------------------------------------------------------------------------
#define MOM_CHUNKS 64

template <typename T>
__global__ __launch_bounds__(256) void mom1_kernel(const T* __restrict__ x, const int64_t* __restrict__ idx,
                                                   const double* __restrict__ w, int64_t n, int D,
                                                   double* __restrict__ part /* [MOM_CHUNKS][D + 2] */) {
    const int c = blockIdx.x;
    const int64_t per = (n + MOM_CHUNKS - 1) / MOM_CHUNKS;
    const int64_t lo = c * per, hi = lo + per < n ? lo + per : n;
    for (int j = threadIdx.x; j < D + 2; j += 256) {
        double s = 0.0;
        for (int64_t r = lo; r < hi; ++r) {
            const double wr = w ? w[r] : 1.0;
            if (j < D) { const int64_t row = idx ? idx[r] : r; s += wr * (double)x[row * D + j]; }
            else if (j == D) s += wr;
            else s += wr * wr;
        }
        part[(size_t)c * (D + 2) + j] = s;
    }
}

__global__ __launch_bounds__(256) void mom1_final_kernel(const double* __restrict__ part, int D, double* __restrict__ mean,
                                                         double* __restrict__ v) {
    __shared__ double v1;
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int c = 0; c < MOM_CHUNKS; ++c) { a += part[(size_t)c * (D + 2) + D]; b += part[(size_t)c * (D + 2) + D + 1]; }
        v[0] = a; v[1] = b; v1 = a;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < D; j += 256) {
        double s = 0.0;
        for (int c = 0; c < MOM_CHUNKS; ++c) s += part[(size_t)c * (D + 2) + j];
        mean[j] = s / v1;
    }
}

// grid (MOM_CHUNKS, tiles_i * tiles_j): a 16 x 16 tile of S over one row chunk
template <typename T>
__global__ __launch_bounds__(256) void mom2_kernel(const T* __restrict__ x, const int64_t* __restrict__ idx,
                                                   const double* __restrict__ w, const double* __restrict__ mean, int64_t n,
                                                   int D, double* __restrict__ part /* [MOM_CHUNKS][D][D] */) {
    __shared__ double xi[16][17], xj[16][17], wr[16];
    const int c = blockIdx.x, nt = (D + 15) / 16;
    const int ti = blockIdx.y / nt, tj = blockIdx.y % nt;
    if (tj < ti) return;                                            // symmetric: upper tiles only
    const int a = threadIdx.x >> 4, b = threadIdx.x & 15;
    const int i = 16 * ti + a, j = 16 * tj + b;
    const int64_t per = (n + MOM_CHUNKS - 1) / MOM_CHUNKS;
    const int64_t lo = c * per, hi = lo + per < n ? lo + per : n;
    double s = 0.0;
    for (int64_t r0 = lo; r0 < hi; r0 += 16) {
        {   // 16 rows x 16 columns of both tiles, centred
            const int64_t r = r0 + a;
            double vi = 0.0, vj = 0.0;
            if (r < hi) {
                const int64_t row = idx ? idx[r] : r;
                const int ci = 16 * ti + b, cj = 16 * tj + b;
                if (ci < D) vi = (double)x[row * D + ci] - mean[ci];
                if (cj < D) vj = (double)x[row * D + cj] - mean[cj];
                if (b == 0) wr[a] = w ? w[r] : 1.0;
            } else if (b == 0) wr[a] = 0.0;
            xi[a][b] = vi; xj[a][b] = vj;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) s += (wr[k] * xi[k][a]) * xj[k][b];
        __syncthreads();
    }
    if (i < D && j < D) part[((size_t)c * D + i) * D + j] = s;
}

__global__ __launch_bounds__(256) void mom2_final_kernel(const double* __restrict__ part, int D, double* __restrict__ S) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < D * D; e += gridDim.x * 256) {
        const int i = e / D, j = e % D;
        const int ii = i <= j ? i : j, jj = i <= j ? j : i;           // the upper triangle was computed
        const int ti = ii >> 4, tj = jj >> 4;
        (void)ti; (void)tj;
        double s = 0.0;
        for (int c = 0; c < MOM_CHUNKS; ++c) s += part[((size_t)c * D + ii) * D + jj];
        S[e] = s;
    }
}

extern "C" int64_t pmc_moments_workspace_bytes(int32_t D) {
    return (int64_t)sizeof(double) * ((int64_t)MOM_CHUNKS * (D + 2) + (int64_t)MOM_CHUNKS * D * D) + 256;
}

// x: f64 [*][D] (x32 == NULL) or f32 (x32 != NULL); idx i64 [n] or NULL (rows 0..n-1); w f64 [n] or NULL
extern "C" int pmc_moments(const double* x, const float* x32, const int64_t* idx, const double* w, int64_t n, int32_t D,
                           double* mean, double* S, double* v, void* workspace, int64_t workspace_bytes, void* stream) {
    if ((!x && !x32) || !mean || !S || !v || !workspace || n < 1 || D < 1) return pmc_fail("pmc_moments: bad argument");
    if (workspace_bytes < pmc_moments_workspace_bytes(D)) return pmc_fail("pmc_moments: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    double* p1 = (double*)workspace;
    double* p2 = p1 + (size_t)MOM_CHUNKS * (D + 2);
    const int nt = (D + 15) / 16;
    if (hipMemsetAsync(p2, 0, sizeof(double) * (size_t)MOM_CHUNKS * D * D, st) != hipSuccess) return pmc_fail("pmc_moments: memset");
    if (x32) {
        hipLaunchKernelGGL(mom1_kernel<float>, dim3(MOM_CHUNKS), dim3(256), 0, st, x32, idx, w, n, (int)D, p1);
        hipLaunchKernelGGL(mom1_final_kernel, dim3(1), dim3(256), 0, st, (const double*)p1, (int)D, mean, v);
        hipLaunchKernelGGL(mom2_kernel<float>, dim3(MOM_CHUNKS, nt * nt), dim3(256), 0, st, x32, idx, w, (const double*)mean, n, (int)D, p2);
    } else {
        hipLaunchKernelGGL(mom1_kernel<double>, dim3(MOM_CHUNKS), dim3(256), 0, st, x, idx, w, n, (int)D, p1);
        hipLaunchKernelGGL(mom1_final_kernel, dim3(1), dim3(256), 0, st, (const double*)p1, (int)D, mean, v);
        hipLaunchKernelGGL(mom2_kernel<double>, dim3(MOM_CHUNKS, nt * nt), dim3(256), 0, st, x, idx, w, (const double*)mean, n, (int)D, p2);
    }
    hipLaunchKernelGGL(mom2_final_kernel, dim3((D * D + 255) / 256), dim3(256), 0, st, (const double*)p2, (int)D, S);
    return pmc_check_launch("pmc_moments");
}

// ---------------------------------------------------------------------------------------------------------------
// np.median(data, 1) of student.py:45: per-column median of rows x[idx[r]] -- transpose, one segmented radix sort,
// the middle element (or the mean of the two middle ones, in the input's precision like np.mean)
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_rows_kernel(const T* __restrict__ x, const int64_t* __restrict__ idx, int64_t n,
                                                             int D, T* __restrict__ xt /* [D][n] */) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n * D; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / D; const int j = (int)(e % D);
        const int64_t row = idx ? idx[r] : r;
        xt[(size_t)j * n + r] = x[row * D + j];
    }
}

__global__ void segment_offsets_kernel(int64_t n, int D, int* __restrict__ off) {
    for (int j = threadIdx.x; j <= D; j += blockDim.x) off[j] = (int)(j * n);
}

template <typename T>
__global__ void pick_median_kernel(const T* __restrict__ sorted, int64_t n, int D, T* __restrict__ med) {
    for (int j = threadIdx.x; j < D; j += blockDim.x) {
        const T* s = sorted + (size_t)j * n;
        med[j] = (n & 1) ? s[(n - 1) / 2] : (T)((s[n / 2 - 1] + s[n / 2]) / (T)2);
    }
}

template <typename T>
static int64_t medians_bytes(int64_t n, int32_t D) {
    size_t tmp = 0;
    (void)rocprim::segmented_radix_sort_keys(nullptr, tmp, (const T*)nullptr, (T*)nullptr, (unsigned)(n * D), (unsigned)D,
                                             (const int*)nullptr, (const int*)nullptr);
    return (int64_t)(2 * sizeof(T) * (size_t)n * D + sizeof(int) * (size_t)(D + 1) + tmp + 512);
}

extern "C" int64_t pmc_column_medians_workspace_bytes(int64_t n, int32_t D, int32_t is_f32) {
    return is_f32 ? medians_bytes<float>(n, D) : medians_bytes<double>(n, D);
}

template <typename T>
static int medians_run(const T* x, const int64_t* idx, int64_t n, int32_t D, T* med, void* workspace, int64_t bytes, hipStream_t st) {
    if (n * (int64_t)D > 0x7fffffffLL) return pmc_fail("pmc_column_medians: more than 2^31 elements");
    T* xt = (T*)workspace;
    T* srt = xt + (size_t)n * D;
    int* off = (int*)(srt + (size_t)n * D);
    void* tmp = (void*)(((uintptr_t)(off + D + 1) + 255) & ~(uintptr_t)255);
    size_t need = 0;
    (void)rocprim::segmented_radix_sort_keys(nullptr, need, (const T*)xt, srt, (unsigned)(n * D), (unsigned)D, (const int*)off, (const int*)(off + 1));
    if ((char*)tmp + need > (char*)workspace + bytes) return pmc_fail("pmc_column_medians: workspace too small");
    int64_t grid = (n * D + 255) / 256; if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(transpose_rows_kernel<T>, dim3((unsigned)grid), dim3(256), 0, st, x, idx, n, (int)D, xt);
    hipLaunchKernelGGL(segment_offsets_kernel, dim3(1), dim3(256), 0, st, n, (int)D, off);
    if (rocprim::segmented_radix_sort_keys(tmp, need, (const T*)xt, srt, (unsigned)(n * D), (unsigned)D, (const int*)off,
                                           (const int*)(off + 1), 0u, (unsigned)sizeof(T) * 8, st) != hipSuccess)
        return pmc_fail("pmc_column_medians: sort failed");
    hipLaunchKernelGGL(pick_median_kernel<T>, dim3(1), dim3(256), 0, st, (const T*)srt, n, (int)D, med);
    return pmc_check_launch("pmc_column_medians");
}

// x f64 (x32 == NULL, med64 out) or f32 (med32 out); idx i64 [n] or NULL
extern "C" int pmc_column_medians(const double* x, const float* x32, const int64_t* idx, int64_t n, int32_t D, double* med64,
                                  float* med32, void* workspace, int64_t workspace_bytes, void* stream) {
    if ((!x && !x32) || n < 1 || D < 1 || !workspace) return pmc_fail("pmc_column_medians: bad argument");
    if (x32) {
        if (!med32) return pmc_fail("pmc_column_medians: float32 input needs the float32 output");
        return medians_run<float>(x32, idx, n, D, med32, workspace, workspace_bytes, (hipStream_t)stream);
    }
    if (!med64) return pmc_fail("pmc_column_medians: float64 input needs the float64 output");
    return medians_run<double>(x, idx, n, D, med64, workspace, workspace_bytes, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// bootstrap of the evidence estimate, sampler.py:905-911: out[b] = logsumexp(logw[choice(n, n)]) - log(n) for B
// replicates; one workgroup per replicate, indices from Philox keyed by (seed, replicate, draw)
// ---------------------------------------------------------------------------------------------------------------
// replay != NULL: i64 [B][n] the draws of every replicate (np.random.choice(n, n) of sampler.py:908, recorded) instead of Philox
__global__ __launch_bounds__(256) void bootstrap_lse_kernel(const double* __restrict__ logw, int64_t n, const double* __restrict__ stats,
                                                            uint64_t seed, const int64_t* __restrict__ replay, double* __restrict__ out) {
    __shared__ double part[256];
    const double mx = stats[0];
    double s = 0.0;
    for (int64_t e = (int64_t)threadIdx.x * 2; e < n; e += 512) {
        int64_t i0, i1;
        if (replay) {
            i0 = replay[(size_t)blockIdx.x * n + e];
            i1 = e + 1 < n ? replay[(size_t)blockIdx.x * n + e + 1] : 0;
            // a recorded draw outside [0, n) is the caller's error: the replicate becomes NaN instead of a read out of bounds
            if (i0 < 0 || i0 >= n || i1 < 0 || i1 >= n) { s = __builtin_nan(""); i0 = 0; i1 = 0; }
        } else {
            Philox ph(seed, (uint64_t)blockIdx.x, (uint64_t)(e >> 1), 3);
            double u0, u1;
            ph.uniform2(u0, u1);
            i0 = (int64_t)(u0 * (double)n); if (i0 >= n) i0 = n - 1;
            i1 = (int64_t)(u1 * (double)n); if (i1 >= n) i1 = n - 1;
        }
        s += exp(logw[i0] - mx);
        if (e + 1 < n) s += exp(logw[i1] - mx);
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = mx + log(part[0]) - log((double)n);
}

// stats f64 [>= 1] (device): stats[0] = max(logw) (pmc_logw_stats)
extern "C" int pmc_bootstrap_logz(const double* logw, int64_t n, const double* stats, int64_t B, uint64_t seed, double* out,
                                  void* stream) {
    if (!logw || !stats || !out || n < 1 || B < 1) return pmc_fail("pmc_bootstrap_logz: bad argument");
    hipLaunchKernelGGL(bootstrap_lse_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, logw, n, stats, seed,
                       (const int64_t*)nullptr, out);
    return pmc_check_launch("bootstrap_lse_kernel");
}

// the same with the draws given (parity tests: the reference's recorded np.random.choice draws, sampler.py:908)
extern "C" int pmc_bootstrap_logz_replay(const double* logw, int64_t n, const double* stats, int64_t B, const int64_t* draws,
                                         double* out, void* stream) {
    if (!logw || !stats || !out || !draws || n < 1 || B < 1) return pmc_fail("pmc_bootstrap_logz_replay: bad argument");
    hipLaunchKernelGGL(bootstrap_lse_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, logw, n, stats, (uint64_t)0,
                       draws, out);
    return pmc_check_launch("bootstrap_lse_kernel");
}

// ---------------------------------------------------------------------------------------------------------------
// Full affine map of Reparameterize(diagonal=False), scaler.py:288-292 / :308-313, row by row:
//   mode 0: out = mu + M in   (M = L, the Cholesky factor of cov(u))      mode 1: out = M (in - mu)   (M = L^-1)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void affine_rows_kernel(const double* __restrict__ M, const double* __restrict__ mu,
                                                          const double* __restrict__ in, double* __restrict__ out,
                                                          int64_t n, int D, int mode) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n * D; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / D; const int i = (int)(e % D);
        const double* row = in + r * D;
        double s = 0.0;
        if (mode == 0) { for (int j = 0; j < D; ++j) s += M[i * D + j] * row[j]; s = mu[i] + s; }
        else { for (int j = 0; j < D; ++j) s += M[i * D + j] * (row[j] - mu[j]); }
        out[e] = s;
    }
}

extern "C" int pmc_affine_rows(const double* M, const double* mu, const double* in, double* out, int64_t n, int32_t D,
                               int32_t mode, void* stream) {
    if (!M || !mu || !in || !out || in == out || n < 1 || D < 1 || (mode != 0 && mode != 1))
        return pmc_fail("pmc_affine_rows: bad argument");
    int64_t grid = (n * D + 255) / 256; if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(affine_rows_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, M, mu, in, out, n, (int)D, (int)mode);
    return pmc_check_launch("affine_rows_kernel");
}

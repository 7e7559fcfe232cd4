// trim_weights (pocomc/tools.py:10-53): drop the lowest-weight samples while the effective sample
// size stays >= ess * ESS_total, scanning `bins` percentile thresholds downward from the 99th.
//
// The reference evaluates np.percentile (a partition) and two masked sums per threshold -- up to
// 1000 passes over the pool.  Here: one radix sort (rocPRIM via hipCUB), two prefix sums, and one
// 1024-thread kernel that evaluates every threshold at once and picks the first one the
// reference's downward scan would accept.

#include <hip/hip_runtime.h>
#include <string.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>
#include "pmc_internal.h"

#pragma clang fp contract(off)

__global__ __launch_bounds__(256) void square_kernel(const double* __restrict__ a, double* __restrict__ b, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) b[e] = a[e] * a[e];
}

// sorted ascending a[P]; inclusive prefix sums c1 (of a) and c2 (of a^2)
__global__ __launch_bounds__(1024) void trim_search_kernel(const double* __restrict__ a, const double* __restrict__ c1,
                                                           const double* __restrict__ c2, int64_t P, double ess,
                                                           int bins, double* __restrict__ result) {
    __shared__ int best;
    if (threadIdx.x == 0) best = -1;
    __syncthreads();
    const double tot1 = c1[P - 1], tot2 = c2[P - 1];
    const double ess_total = (tot1 * tot1) / tot2;
    for (int i = threadIdx.x; i < bins; i += blockDim.x) {
        // percentiles = np.linspace(0, 99, bins)
        const double step = 99.0 / (double)(bins - 1);
        const double pct = (i == bins - 1) ? 99.0 : (double)i * step;
        // np.percentile, method='linear'
        const double vidx = (double)(P - 1) * (pct / 100.0);
        int64_t lo = (int64_t)floor(vidx);
        if (lo > P - 1) lo = P - 1;
        const int64_t hi = lo + 1 < P ? lo + 1 : P - 1;
        const double g = vidx - (double)lo;
        const double av = a[lo], bv = a[hi], d = bv - av;
        const double thr = (g >= 0.5) ? bv - d * (1.0 - g) : av + d * g;
        // first sorted index with a[k] >= thr
        int64_t l = 0, h = P;
        while (l < h) { const int64_t mid = (l + h) >> 1; if (a[mid] < thr) l = mid + 1; else h = mid; }
        const double s1 = tot1 - (l > 0 ? c1[l - 1] : 0.0), s2 = tot2 - (l > 0 ? c2[l - 1] : 0.0);
        const double ess_t = (s1 * s1) / s2;
        if (l < P && ess_t / ess_total >= ess) atomicMax(&best, i);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int i = best < 0 ? 0 : best;      // the reference's loop always terminates at i = 0 (threshold = min)
        const double step = 99.0 / (double)(bins - 1);
        const double pct = (i == bins - 1) ? 99.0 : (double)i * step;
        const double vidx = (double)(P - 1) * (pct / 100.0);
        int64_t lo = (int64_t)floor(vidx);
        if (lo > P - 1) lo = P - 1;
        const int64_t hi = lo + 1 < P ? lo + 1 : P - 1;
        const double g = vidx - (double)lo;
        const double av = a[lo], bv = a[hi], d = bv - av;
        result[0] = (g >= 0.5) ? bv - d * (1.0 - g) : av + d * g;
        result[1] = (double)i;
    }
}

extern "C" int64_t pmc_trim_workspace_bytes(int64_t P) {
    size_t tmp_sort = 0, tmp_scan = 0;
    (void)rocprim::radix_sort_keys(nullptr, tmp_sort, (const double*)nullptr, (double*)nullptr, (size_t)P);
    (void)rocprim::inclusive_scan(nullptr, tmp_scan, (const double*)nullptr, (double*)nullptr, (size_t)P, rocprim::plus<double>());
    const size_t tmp = tmp_sort > tmp_scan ? tmp_sort : tmp_scan;
    return (int64_t)(4 * (size_t)P * sizeof(double) + tmp + 256);
}

extern "C" int pmc_trim_threshold(const double* w, int64_t P, double ess, int32_t bins, double* result,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
    if (!w || !result || !workspace || P < 1 || bins < 2 || bins > 65536) return pmc_fail("pmc_trim_threshold: bad argument");
    if (workspace_bytes < pmc_trim_workspace_bytes(P)) return pmc_fail("pmc_trim_threshold: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    double* sorted = (double*)workspace;
    double* sq = sorted + P;
    double* c1 = sq + P;
    double* c2 = c1 + P;
    void* tmp = (void*)(c2 + P);
    size_t tmp_bytes = (size_t)workspace_bytes - 4 * (size_t)P * sizeof(double);
    size_t need = 0;
    (void)rocprim::radix_sort_keys(nullptr, need, w, sorted, (size_t)P);
    if (rocprim::radix_sort_keys(tmp, need, w, sorted, (size_t)P, 0u, 64u, st) != hipSuccess)
        return pmc_fail("pmc_trim_threshold: radix sort failed");
    int64_t grid = (P + 255) / 256; if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(square_kernel, dim3((unsigned)grid), dim3(256), 0, st, (const double*)sorted, sq, P);
    (void)rocprim::inclusive_scan(nullptr, need, (const double*)sorted, c1, (size_t)P, rocprim::plus<double>());
    if (need > tmp_bytes) return pmc_fail("pmc_trim_threshold: workspace too small (scan)");
    if (rocprim::inclusive_scan(tmp, need, (const double*)sorted, c1, (size_t)P, rocprim::plus<double>(), st) != hipSuccess)
        return pmc_fail("pmc_trim_threshold: scan failed");
    if (rocprim::inclusive_scan(tmp, need, (const double*)sq, c2, (size_t)P, rocprim::plus<double>(), st) != hipSuccess)
        return pmc_fail("pmc_trim_threshold: scan failed");
    hipLaunchKernelGGL(trim_search_kernel, dim3(1), dim3(1024), 0, st, (const double*)sorted, (const double*)c1,
                       (const double*)c2, P, ess, (int)bins, result);
    return pmc_check_launch("trim_search_kernel");
}

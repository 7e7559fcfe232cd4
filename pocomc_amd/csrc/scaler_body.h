// Element-wise pieces of the scaler (pocomc/scaler.py) and of Prior.logpdf (pocomc/prior.py), shared by the scaler
// kernels (mcmc_kernels.hip) and by the epilogue of the flow-inverse sweeps (maf_inverse_tri4.hip), which applies the
// scaler to its 16 walkers while they are still in LDS: one launch and one global round trip less per MCMC step.
// float64 in numpy's operation order, no FMA contraction: both users produce the same bits.
#ifndef PMC_SCALER_BODY_H
#define PMC_SCALER_BODY_H

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/pocomc_amd.h"

#pragma clang fp contract(off)

// ===========================================================================
// scaler: Reparameterize.inverse / forward (scaler.py:180-226, :293-425)
// ===========================================================================
#define LOG_SQRT_2PI 0.91893853320467267   // np.log(np.sqrt(2.0*np.pi))
#define SQRT2 1.4142135623730951           // np.sqrt(2.0)

// numpy's pairwise summation (umath loops, PW_BLOCKSIZE = 128), so that the row sum of
// the Jacobian terms (scaler.py:270) is accumulated in numpy's order
__device__ __forceinline__ double np_pairwise_leaf(const double* a, int n) {   // n <= 128
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += a[i + 0]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += a[i];
    return res;
}

// numpy splits n > 128 in halves (first half rounded down to a multiple of 8), recursively;
// three explicit levels cover n <= 1024 without a device-side call stack
template <int LEVEL>
__device__ __forceinline__ double np_pairwise_sum_l(const double* a, int n) {
    if (n <= 128) return np_pairwise_leaf(a, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    if constexpr (LEVEL > 0) return np_pairwise_sum_l<LEVEL - 1>(a, n2) + np_pairwise_sum_l<LEVEL - 1>(a + n2, n - n2);
    else return np_pairwise_leaf(a, n2) + np_pairwise_leaf(a + n2, n - n2);
}

__device__ __forceinline__ double np_pairwise_sum(const double* a, int n) { return np_pairwise_sum_l<2>(a, n); }

// bounded <- unbounded for one element; t is the (affine-transformed) input.  scaler.py:329-425
__device__ __forceinline__ void bound_inverse(const pmc_scaler_t& s, int j, double t, double& x, double& J) {
    {
        const int kind = s.kind[j];
        if (kind == 0) { x = t; J = 0.0; }
        else if (kind == 1) { x = exp(t) + s.low[j]; J = t; }
        else if (kind == 2) { x = s.high[j] - exp(t); J = t; }
        else {
            const double w = s.high[j] - s.low[j];
            if (s.logit) {
                // p = exp(-logaddexp(0, -t))
                const double mt = -t;
                double lae;
                if (mt == 0.0) lae = 0.6931471805599453;
                else if (0.0 - mt > 0.0) lae = 0.0 + log1p(exp(-(0.0 - mt)));
                else lae = mt + log1p(exp(0.0 - mt));
                const double p = exp(-lae);
                x = p * w + s.low[j];
                J = (s.log_width[j] + log(p)) + log(1.0 - p);
            } else {
                const double p = (erf(t / SQRT2) + 1.0) / 2.0;
                x = p * w + s.low[j];
                J = (s.log_width[j] + (-(t * t) / 2.0)) - LOG_SQRT_2PI;
            }
        }
    }
}

// unbounded <- bounded (scaler.py:228-247, :315-400), then the affine part (:273-289)
__device__ __forceinline__ double bound_forward(const pmc_scaler_t& s, int j, double x) {
    double u;
    {
        const int kind = s.kind[j];
        if (kind == 0) u = x;
        else if (kind == 1) u = log(x - s.low[j]);
        else if (kind == 2) u = log(s.high[j] - x);
        else {
            const double p = (x - s.low[j]) / (s.high[j] - s.low[j]);
            if (s.logit) u = log(p / (1.0 - p));
            else u = SQRT2 * erfinv(2.0 * p - 1.0);
        }
        if (s.scale) u = (u - s.mu[j]) / s.sigma[j];
    }
    return u;
}

// periodic wrap / reflective fold (scaler.py:109-157).  The reference loops "while
// outside"; a non-finite x would never leave that loop, the device bounds it.
__device__ __forceinline__ double apply_bc(const pmc_scaler_t& s, int j, double x) {
    const int bc = s.bc[j];
    if (bc == 0) return x;
    const double lo = s.low[j], hi = s.high[j];
    {
        if (bc & 1) {
            for (int it = 0; it < 4096 && x > hi; ++it) x = lo + x - hi;
            for (int it = 0; it < 4096 && x < lo; ++it) x = hi + x - lo;
        }
        if (bc & 2) {
            for (int it = 0; it < 4096 && x > hi; ++it) x = hi - x + hi;
            for (int it = 0; it < 4096 && x < lo; ++it) x = lo + lo - x;
        }
    }
    return x;
}

// one factor of Prior.logpdf (pocomc/prior.py:70-100), shared by prior_logpdf_kernel and the fused scaler kernel
__device__ __forceinline__ double prior_term(const pmc_prior_t& pr, int j, double xv) {
    const double loc = pr.loc[j], sc = pr.scale[j];
    if (pr.family[j] == PMC_PRIOR_UNIFORM) {
        // scipy uniform(loc, scale).logpdf: -log(scale) on [loc, loc+scale], -inf outside
        return (xv >= loc && xv <= loc + sc) ? -log(sc) : -INFINITY;
    }
    // scipy norm(loc, scale).logpdf: _norm_logpdf((x-loc)/scale) - log(scale)
    const double z = (xv - loc) / sc;
    return (-(z * z) / 2.0 - LOG_SQRT_2PI) - log(sc);
}


// ---------------------------------------------------------------------------------------------------------------
// The scaler (+ prior) as the epilogue of a sweep: what scaler_inverse_kernel does for 64 rows, for the 16 walkers
// of one workgroup.  X: the sweep's result in LDS, by rank of the first transform ([rank][16], lidx); rof: rank of
// every feature; scr: LDS scratch of scaler_epilogue_lds_bytes(D) bytes, 16-byte aligned.  Every thread of the
// workgroup calls it (it synchronises).
struct ScalerEpi {
    int on, have_prior;
    pmc_scaler_t s;
    pmc_prior_t pr;
    double* u_out; double* x_out; double* x_colmajor; double* ldj_out; int32_t* finite_out;
    double* logp_out; int32_t* finite_copy; double* logp_copy;
    unsigned* done_ticket; long long* done_flag; long long done_value;
    unsigned* bad_count; long long* bad_flag;       // optional: rows that are not clean (pmc_step_t.h_clean)
    const double* fill_x;                           // pmc_step_t.fill_rejected: the walkers' current x (device [n][D]) or NULL
    long long* stamps;                              // measurement only (pmc_debug_set_epilogue_stamps): [block][8] of the 100 MHz clock
};

static inline size_t scaler_epilogue_lds_bytes(int D) {
    return (size_t)(2 * 16 * D + D * 17) * sizeof(double) + 16 * sizeof(int);
}

// one element (walker r of the workgroup's 16, feature j) of the epilogue: the bijector, its Jacobian term, the boundary
// conditions, the prior factor; u / x to the device arrays, x to the host's column-major array when asked (direct_cm),
// the row tables Jt / Pt / rowfin (and Xt unless direct_cm) in LDS
__device__ __forceinline__ void scaler_epilogue_element(const ScalerEpi& e, double* Jt, double* Pt, double* Xt, int* rowfin,
                                                        float u32, int r, int j, int64_t row0, int64_t n, int D, bool direct_cm) {
    const pmc_scaler_t& s = e.s;
    const int64_t g = (row0 + r) * D + j;
    double u = (double)u32;
    double t, x, J;
    t = s.scale ? s.mu[j] + s.sigma[j] * u : u;
    bound_inverse(s, j, t, x, J);
    if (s.bc) {
        // mcmc.py:94-97: wrap x, re-derive u from it, invert again
        x = apply_bc(s, j, x);
        u = bound_forward(s, j, x);
        t = s.scale ? s.mu[j] + s.sigma[j] * u : u;
        bound_inverse(s, j, t, x, J);
    }
    e.u_out[g] = u;
    e.x_out[g] = x;
    if (direct_cm) { if (e.x_colmajor) e.x_colmajor[(size_t)j * n + row0 + r] = x; }
    else Xt[j * 17 + r] = x;
    Jt[r * D + j] = J;
    if (e.have_prior) Pt[r * D + j] = prior_term(e.pr, j, x);
    if (!isfinite(x)) rowfin[r] = 0;
}

// the rows' part: log-determinant (numpy's pairwise sum of the Jacobian terms), finite mask, Prior.logpdf; then the
// column-major store of x (unless the elements stored it already: cm_done) and the completion word.  Every thread of
// the workgroup calls it behind a barrier that follows the last element.
__device__ __forceinline__ void scaler_epilogue_rows(const ScalerEpi& e, const double* Jt, const double* Pt, const double* Xt,
                                                     int* rowfin, int64_t row0, int64_t n, int D, int tid, int nthr,
                                                     bool cm_done) {
    const pmc_scaler_t& s = e.s;
    const int rows = (int)min((int64_t)16, n - row0);
    if (tid < rows) {
        double l = np_pairwise_sum(Jt + (size_t)tid * D, D);
        if (s.scale) l = s.sum_log_sigma + l;
        e.ldj_out[row0 + tid] = l;
        const int fin = (rowfin[tid] && isfinite(l)) ? 1 : 0;
        e.finite_out[row0 + tid] = fin;
        if (e.finite_copy) e.finite_copy[row0 + tid] = fin;
        bool clean = fin != 0;
        if (e.have_prior) {
            // Prior.logpdf of the finite rows (mcmc.py:105-107): the terms dimension after dimension
            double lp = -INFINITY;
            if (fin) {
                lp = 0.0;
                for (int j = 0; j < D; ++j) lp += Pt[tid * D + j];
            }
            e.logp_out[row0 + tid] = lp;
            if (e.logp_copy) e.logp_copy[row0 + tid] = lp;
            clean = clean && isfinite(lp);
        }
        if (e.bad_count && !clean) atomicAdd(e.bad_count, 1u);            // (rows that the host's masks would drop: rare)
        if (e.fill_x) rowfin[tid] = clean ? 1 : 0;                              // (from here on: "the row reaches the likelihood")
    }
    if (e.fill_x) __syncthreads();
    if (e.stamps && tid == 0) e.stamps[blockIdx.x * 8 + 2] = wall_clock64();
    if (e.stamps && tid == 0) e.stamps[blockIdx.x * 8 + 3] = wall_clock64();
    if (e.x_colmajor && !cm_done) {
        for (int el = tid; el < rows * D; el += nthr) {
            const int j = el / rows, r = el - j * rows;
            // (a row that does not reach the likelihood: the walker's current x in the host copy, see scaler_inverse_kernel)
            e.x_colmajor[(size_t)j * n + row0 + r] = (e.fill_x && !rowfin[r]) ? e.fill_x[(row0 + r) * D + j] : Xt[j * 17 + r];
        }
    }
    if (e.stamps && tid == 0) e.stamps[blockIdx.x * 8 + 4] = wall_clock64();
    if (e.done_flag) {
        // Every thread waits for the acknowledgement of its own stores (agent-scope release: pinned host memory is not
        // cached on the device, so there is nothing to write back), the workgroup draws a ticket (agent-scope acq_rel), the
        // last one publishes the word with a system-scope release.  A __threadfence_system() here made every one of the ~400
        // workgroups write back and invalidate the whole L2: 12 us of the launch (timing-only builds of this file), 6 us in
        // scripts/micro/pcie_store.hip, which also checks that the host never sees the word before the data (0 stale words
        // in 8000 launches x 208 k words with either fence).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (e.stamps && tid == 0) e.stamps[blockIdx.x * 8 + 5] = wall_clock64();
        if (tid == 0) {
            const unsigned t = __hip_atomic_fetch_add(e.done_ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (t == gridDim.x - 1) {
                *e.done_ticket = 0u;
                if (e.bad_count && e.bad_flag) {          // (every block's count is in: its ticket came behind its atomicAdd)
                    const unsigned bad = __hip_atomic_exchange(e.bad_count, 0u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                    *e.bad_flag = (long long)bad;
                }
                __threadfence_system();
                __hip_atomic_store(e.done_flag, e.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

template <class LIDX>
__device__ __forceinline__ void scaler_epilogue(const ScalerEpi& e, const float* X, const int* __restrict__ rof,
                                                double* scr, int64_t row0, int64_t n, int D, int tid, int nthr,
                                                LIDX lidx_of) {
    double* Jt = scr;                               // [16][D]
    double* Pt = Jt + 16 * D;                       // [16][D] prior terms
    double* Xt = Pt + 16 * D;                       // [D][17]
    int* rowfin = reinterpret_cast<int*>(Xt + D * 17);
    const int rows = (int)min((int64_t)16, n - row0);
    if (tid < 16) rowfin[tid] = 1;
    if (e.stamps && tid == 0) e.stamps[blockIdx.x * 8 + 0] = wall_clock64();
    __syncthreads();
    for (int el = tid; el < rows * D; el += nthr) {
        const int r = el / D, j = el - r * D;
        scaler_epilogue_element(e, Jt, Pt, Xt, rowfin, X[lidx_of(rof[j], r)], r, j, row0, n, D, false);
    }
    __syncthreads();
    if (e.stamps && tid == 0) e.stamps[blockIdx.x * 8 + 1] = wall_clock64();
    scaler_epilogue_rows(e, Jt, Pt, Xt, rowfin, row0, n, D, tid, nthr, false);
}

// the progressive form's tables only (no Xt): [16][D] Jacobian terms, [16][D] prior terms, 16 row flags
static inline size_t scaler_progressive_lds_bytes(int D) { return (size_t)(2 * 16 * D) * sizeof(double) + 16 * sizeof(int); }

#endif

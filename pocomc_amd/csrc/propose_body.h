// Proposal of one wavefront's 16 walkers on the f64 matrix cores (pocomc/mcmc.py:77-85, :251-253), shared by
// propose_mfma_kernel and the flow-inverse kernel that runs it as its prologue (maf_inverse_tri4.hip).
//
// The three D x D products per walker of the tpCN proposal
//     delta  = diff^T  S diff        (S = inv_cov, mcmc.py:80)
//     L z                            (L = chol_cov, mcmc.py:85)
//     delta' = diff'^T S diff'       (mcmc.py:127-128)
// are batched over the 16 walkers as  S . diff^T,  L . z^T,  S . diff'^T  with
// v_mfma_f64_16x16x4_f64 (A = matrix tile [16 i x 4 j], B = walker slice [4 j x 16 walkers]).
// Lane (q = lane>>4, p = lane&15) keeps coordinates {q + 4m} of walker p in registers: that set
// is at once the B operand of every K step, the rows the lane receives in the f64 C layout
// (row = q + 4*reg), and the coordinates of theta' it produces -- so the proposal, its quadratic
// form and the stores never leave the lane; only the two dot products need a 4-quad shuffle sum.
//
// float64 like the reference; the expressions of mcmc.py:80/:85 keep their operation order
// (no FMA contraction), the D-term sums inside the products are accumulated by the MFMA.
#ifndef PMC_PROPOSE_BODY_H
#define PMC_PROPOSE_BODY_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "philox.h"
#include "pmc_internal.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));
#define DMFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ double quad_sum_d(double v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// acc[it][r] = sum_j MAT[16*it + q + 4r][j] * vec[j]   for the lane's walker
// LOWER: MAT is lower-triangular, K steps above the diagonal tile are skipped.
// The lane's A operands of a whole matrix (frag[it][k] = MAT[16 it + (lane & 15)][4 k + q]) requested ahead of the
// products: small proposals (M <= 8) keep both matrices in registers from the start of the kernel -- the loads are in
// flight while the variates are generated, and the second quadratic form reuses S.  Same products in the same order.
template <int M, bool LOWER>
__device__ __forceinline__ void matfrag16(const double* __restrict__ mat, int D, double (&frag)[M / 4][M], int q, int p_lane) {
#pragma unroll
    for (int it = 0; it < M / 4; ++it) {
        const int i = 16 * it + p_lane;
#pragma unroll
        for (int k = 0; k < M; ++k) {
            const int j = 4 * k + q;
            frag[it][k] = (16 * it < D && 4 * k < D && (!LOWER || 4 * k <= 16 * it + 15) && i < D && j < D) ? mat[(size_t)i * D + j] : 0.0;
        }
    }
}
template <int M, bool LOWER>
__device__ __forceinline__ void matvec16_frag(const double (&frag)[M / 4][M], int D, const double (&vec)[M], double (&res)[M]) {
#pragma unroll
    for (int it = 0; it < M / 4; ++it) {
        if (16 * it < D) {
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < M; ++k)
                if (4 * k < D && (!LOWER || 4 * k <= 16 * it + 15)) acc = DMFMA(frag[it][k], vec[k], acc);
            res[4 * it + 0] = acc[0]; res[4 * it + 1] = acc[1]; res[4 * it + 2] = acc[2]; res[4 * it + 3] = acc[3];
        }
    }
}
template <int M, bool LOWER>
__device__ __forceinline__ void matvec16(const double* __restrict__ mat, int D, const double (&vec)[M],
                                         double (&res)[M], int q, int p_lane) {
    const int i_l = p_lane;              // A operand: row within tile = lane & 15
#pragma unroll
    for (int it = 0; it < M / 4; ++it) {
        if (16 * it < D) {
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            const int i = 16 * it + i_l;
            // (the row tile's operands are all requested before the first product: one at a time, a wide proposal --
            // D = 128: 640 products per wavefront -- paid an L2 round trip per product; same products, same order)
            double a[M];
#pragma unroll
            for (int k = 0; k < M; ++k) {
                const int j = 4 * k + q;
                a[k] = (4 * k < D && (!LOWER || 4 * k <= 16 * it + 15) && i < D && j < D) ? mat[(size_t)i * D + j] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < M; ++k)
                if (4 * k < D && (!LOWER || 4 * k <= 16 * it + 15)) acc = DMFMA(a[k], vec[k], acc);
            res[4 * it + 0] = acc[0]; res[4 * it + 1] = acc[1]; res[4 * it + 2] = acc[2]; res[4 * it + 3] = acc[3];
        }
    }
}

// y_lds != NULL: theta' is also written as float32 into an LDS array indexed [rank_of_feat[j]][walker] in the
// MFMA operand layout of the flow kernels (the input of the inverse sweep that follows in the same wave).
template <int M>
__device__ __forceinline__ void propose_body(
    int kind, const float* __restrict__ cur32, const double* __restrict__ cur64,
    const double* __restrict__ mu, const double* __restrict__ inv_cov, const double* __restrict__ chol,
    double nu, double sigma, double cn_a, const pmc_rng_t& rng, double* __restrict__ prop64,
    float* __restrict__ prop32, double* __restrict__ quad, double* __restrict__ quad_prop,
    int64_t n, int D, float* y_lds, const int* __restrict__ rank_of_feat, int64_t set = -1) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, p = lane & 15;
    const int64_t k_w = (set < 0 ? (int64_t)blockIdx.x : set) * 16 + p;    // this lane's walker (set: 16-walker set index)
    const bool live = k_w < n;
    const int64_t gidx = rng.offset + k_w;
    const bool tpcn = (kind == PMC_KIND_TPCN);

    // coordinates j = q + 4m of the walker: current position (minus mu for tpCN), noise
    constexpr bool FRAG = M <= 8;                         // both matrices in registers (2 x M^2 / 4 doubles per lane)
    double fS[FRAG ? M / 4 : 1][FRAG ? M : 1], fL[FRAG ? M / 4 : 1][FRAG ? M : 1];
    if constexpr (FRAG) {
        if (tpcn) matfrag16<M, false>(inv_cov, D, fS, q, p);
        matfrag16<M, true>(chol, D, fL, q, p);
    }
    double dif[M], zz[M], muv[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int j = q + 4 * m;
        double v = 0.0, z = 0.0, mj = 0.0;
        if (j < D) {
            mj = tpcn ? mu[j] : 0.0;
            if (live) {
                v = cur32 ? (double)cur32[k_w * D + j] : cur64[k_w * D + j];
                if (rng.normal) z = rng.normal[k_w * D + j];
            }
        }
        muv[m] = mj;
        dif[m] = tpcn ? v - mj : v;
        zz[m] = z;
    }
    if (!rng.normal) {
        // Box-Muller pair (j >> 1) serves coordinates j and j ^ 1, which lanes q and q ^ 1 of the walker hold at the same
        // m.  Of every two m the lane generates ONE pair -- m = 2 i + (q & 1) -- keeps its half and hands the partner the
        // other (one double over the 16-lane distance); the partner does the same for the other m: half the Philox rounds
        // and transcendentals, every lane busy in every round, same values.
        static_assert(M % 2 == 0, "coordinates come in pairs of m");
        const int odd = q & 1;
#pragma unroll
        for (int i = 0; i < M / 2; ++i) {
            const int j = q + 4 * (2 * i + odd);
            double a = 0.0, b = 0.0;
            if ((j & ~1) < D && live) {                       // (the pair's even member exists)
                Philox ph(rng.seed, rng.step, gidx, 1);
                ph.ctr[0] = (uint32_t)(j >> 1);              // pair index: same stream as a sequential walk
                ph.normal2(a, b);
            }
            const double own = odd ? b : a, got = __shfl_xor(odd ? a : b, 16);
            const double z0 = odd ? got : own, z1 = odd ? own : got;     // m = 2 i and m = 2 i + 1
            if (q + 4 * (2 * i) < D && live) zz[2 * i] = z0;
            if (q + 4 * (2 * i + 1) < D && live) zz[2 * i + 1] = z1;
        }
    }

    double scale_z = sigma, q_cur = 0.0;
    double tmp[M];
    if (tpcn) {
        if constexpr (FRAG) matvec16_frag<M, false>(fS, D, dif, tmp);
        else matvec16<M, false>(inv_cov, D, dif, tmp, q, p);         // rows q+4m of S.diff
        double part = 0.0;
#pragma unroll
        for (int m = 0; m < M; ++m) part += dif[m] * tmp[m];
        q_cur = quad_sum_d(part);
        double g;
        if (rng.gamma) g = live ? rng.gamma[k_w] : 1.0;
        else { Philox ph(rng.seed, rng.step, gidx, 0); g = ph.std_gamma(0.5 * ((double)D + nu)); }
        const double s = 1.0 / ((2.0 / (nu + q_cur)) * g);       // 1/np.random.gamma(shape, scale), mcmc.py:80
        scale_z = sigma * sqrt(s);
    }
    if constexpr (FRAG) matvec16_frag<M, true>(fL, D, zz, tmp);
    else matvec16<M, true>(chol, D, zz, tmp, q, p);                  // rows q+4m of L.z
    double prop[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        if (tpcn) prop[m] = (muv[m] + cn_a * dif[m]) + scale_z * tmp[m];     // mcmc.py:85
        else prop[m] = dif[m] + scale_z * tmp[m];                             // mcmc.py:253
    }
    if (tpcn) {
        double dp[M];
#pragma unroll
        for (int m = 0; m < M; ++m) dp[m] = (q + 4 * m < D) ? prop[m] - muv[m] : 0.0;
        if constexpr (FRAG) matvec16_frag<M, false>(fS, D, dp, tmp);
        else matvec16<M, false>(inv_cov, D, dp, tmp, q, p);
        double part = 0.0;
#pragma unroll
        for (int m = 0; m < M; ++m) part += dp[m] * tmp[m];
        const double q_new = quad_sum_d(part);
        if (live && q == 0) {
            if (quad) quad[k_w] = q_cur;
            if (quad_prop) quad_prop[k_w] = q_new;
        }
    }
    if (live) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int j = q + 4 * m;
            if (j < D) {
                if (prop64) prop64[k_w * D + j] = prop[m];
                if (prop32) prop32[k_w * D + j] = (float)prop[m];
            }
        }
    }
    if (y_lds) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int j = q + 4 * m;
            if (j < D) {
                const int r = rank_of_feat[j];
                y_lds[((r >> 4) << 8) + ((r & 3) << 6) + (p << 2) + ((r >> 2) & 3)] = live ? (float)prop[m] : 0.0f;
            }
        }
    }
}

// Proposal arguments of the fused variant (pmc_propose_inverse): the wave first proposes theta' for its 16
// walkers (propose_body.h) straight into the sweep's LDS input -- one launch and one global round trip less
// per MCMC step.
#include "scaler_body.h"

struct ProposeArgs {
    int kind;
    const float* cur32;
    const double* mu; const double* inv_cov; const double* chol;
    double nu, sigma, cn_a;
    pmc_rng_t rng;
    double* prop64; double* quad; double* quad_prop;
    const double* adapt;          // pmc_step_t.adapt_state or NULL: {sigma, cn_a, mu[D]} on the device
    long long* prof;              // measurement only (scripts/profile_tri6.py): cycle stamps of workgroup 0, or NULL
    ScalerEpi epi;                // epi.on: the scaler (+ prior) runs as the sweep's epilogue (scaler_body.h)
};

// lane-per-walker sweep (maf_inverse_tri6.hip); pa == nullptr: plain inverse of z.  -1: flow not covered
bool pmc_tri6_preferred(const pmc_maf_t* m);     // AUTO takes the lane-per-walker sweep for this flow (maf_inverse_tri6.hip)
int pmc_launch_tri6(const ProposeArgs* pa, const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n,
                    hipStream_t stream);
// two-wave spline sweep (maf_inverse_nsf2.hip) with the fused proposal (+ scaler epilogue: *epi_done); -1: flow not covered
int pmc_launch_propose_inverse_nsf2(ProposeArgs* pa, const ScalerEpi* epi, int* epi_done, const pmc_maf_t* m, float* x, float* ladj,
                                    int64_t n, hipStream_t stream);

#endif

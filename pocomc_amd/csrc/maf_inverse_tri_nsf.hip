// Triangular-sweep inverse of the neural spline flows (pocomc/mcmc.py:88 -> flow.py:116-132 with
// flow = nsf3 | nsf6 | nsf12).
//
// Same sweep as the affine kernels (maf_inverse_tri4.hip): hidden units sorted by autoregressive
// degree make the masked weights block lower-triangular, so instead of zuko's D fixed-point passes
// of the full hyper-network the inverse is ONE pass over the degree groups -- per hidden tile a
// left-looking burst against everything already final, then per degree group the dependent chain
//     h0 -> h1 -> h2 -> 23 spline parameters of the rank -> x_rank = spline^-1(y_rank) -> rank-1 update.
// What changes against the affine sweep is the output hop: the 23 parameters of a rank are rows of
// two private 16-row tiles (packed image section f3i), produced by a left-looking product over the
// final h2 tiles whose weight fragments were fetched into registers while the hidden chain of the
// group ran, exchanged through a 2 KB LDS panel so that every lane holds all 23 values of its row,
// and fed to the spline inverse (rqs.h).
//
// One wavefront owns 16 rows; no barriers inside the sweep (a wave's DS operations execute in order).
#include "maf_common.h"
#include "rqs.h"

#define DIAG4N(ACC, FRAG, ACT)                                            \
    {                                                                     \
        if (j0 <= 0 && 0 <= j1) ACC = MFMA(FRAG.x, ACT[hb + 0], ACC);     \
        if (j0 <= 1 && 1 <= j1) ACC = MFMA(FRAG.y, ACT[hb + 1], ACC);     \
        if (j0 <= 2 && 2 <= j1) ACC = MFMA(FRAG.z, ACT[hb + 2], ACC);     \
        if (j0 <= 3 && 3 <= j1) ACC = MFMA(FRAG.w, ACT[hb + 3], ACC);     \
    }

#define NPX 2     // x tiles prefetched per hidden tile
#define NPK 8     // K tiles prefetched per hidden tile and layer
#define NPO 10    // K tiles of the rank's two output tiles prefetched per group
#ifndef NSF_ABL
#define NSF_ABL 0 // timing-only builds (scripts/abl_nsf.sh)
#endif

// ABL (timing experiments only, wrong results): 1 = no spline solve, 2 = no output product, 4 = no hidden chain,
// 8 = no per-rank output fragment loads
template <int ABL>
__global__ __launch_bounds__(64) void maf_inverse_tri_nsf_kernel(pmc_maf_t m, const float* __restrict__ in,
                                                                 float* __restrict__ out,
                                                                 float* __restrict__ ladj_out, int64_t n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int q = lane >> 4, p = lane & 15;
    const int64_t row0 = (int64_t)blockIdx.x * 16;
    const int D = m.D, Dp = m.Dp, Hp = m.Hp, T = m.T, nT = m.nT, nXT = m.nXT;
    float* Y = smem;                 // input of the transform being inverted, by rank
    float* X = Y + Dp * 16;          // its output, filled rank after rank
    float* H0 = X + Dp * 16;
    float* H1 = H0 + Hp * 16;
    float* H2 = H1 + Hp * 16;
    float* PAR = H2 + Hp * 16;       // [16 rows][32]: the 23 spline parameters of the current rank
    float* TAB = PAR + 16 * 32;      // [16 rows][24]: its x / y knot tables (rqs_inverse_coop)
    const int* feat_of_rank = m.meta + 8;
    const int* rank_of_feat = m.meta + 8 + T * D;
    const int* quad_meta = m.meta + 8 + 2 * T * D;

    load_rows(Y, in, row0, n, D, Dp, feat_of_rank + (T - 1) * D, lane);
    float ladj = 0.0f;               // lanes q == 0 accumulate their row's log-determinant

    for (int t = T - 1; t >= 0; --t) {
        const MafView w = maf_view(m, t);
        {
            float4* z4 = reinterpret_cast<float4*>(X);
            const int n4 = (Dp * 16 + 3 * Hp * 16) >> 2;
            for (int e = lane; e < n4; e += 64) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();

        // solve one rank from its 23 parameters (accumulators of its two output tiles)
#define SOLVE_RANK(G, OA0, OA1)                                                                         \
        {                                                                                               \
            float* pr_ = PAR + (p << 5) + (q << 2);                                                     \
            *reinterpret_cast<float4*>(pr_) = make_float4(OA0[0], OA0[1], OA0[2], OA0[3]);              \
            *reinterpret_cast<float4*>(pr_ + 16) = make_float4(OA1[0], OA1[1], OA1[2], OA1[3]);         \
            WAVE_LDS_FENCE();                                                                           \
            float xv_, l_;                                                                              \
            if (ABL & 1) { xv_ = Y[lidx((G), p)] + PAR[(p << 5)] + PAR[(p << 5) + 22]; l_ = PAR[(p << 5) + 8]; } \
            else rqs_inverse_coop(PAR + (p << 5), TAB + p * 24, q, Y[lidx((G), p)], xv_, l_);           \
            if (q == 0) { X[lidx((G), p)] = xv_; ladj -= l_; }                                          \
            WAVE_LDS_FENCE();                                                                           \
        }

        // ---------------- prefetch registers (filled for tile Tt while tile Tt-1 runs)
        float4 pf0[NPX], pf1[NPK], pf2[NPK], pw0[4];
        float4 pd1, pd2, pb0, pb1, pb2;
        int4 pdg;

#define PREFETCH_N(TT)                                                                                     \
        {                                                                                                  \
            const int TT_ = (TT);                                                                          \
            pdg = *reinterpret_cast<const int4*>(quad_meta + 4 * TT_);                                     \
            pdg.x &= 0xffff; pdg.y &= 0xffff; pdg.z &= 0xffff; pdg.w &= 0xffff;                            \
            const float4* f0_ = w.f0 + ((size_t)TT_ * nXT) * 64 + lane;                                    \
            _Pragma("unroll") for (int i_ = 0; i_ < NPX; ++i_) if (i_ < nXT) pf0[i_] = f0_[i_ * 64];       \
            const float4* f1_ = w.f1 + ((size_t)TT_ * nT) * 64 + lane;                                     \
            const float4* f2_ = w.f2 + ((size_t)TT_ * nT) * 64 + lane;                                     \
            _Pragma("unroll") for (int i_ = 0; i_ < NPK; ++i_) if (i_ < TT_) { pf1[i_] = f1_[i_ * 64]; pf2[i_] = f2_[i_ * 64]; } \
            pd1 = f1_[TT_ * 64]; pd2 = f2_[TT_ * 64];                                                      \
            const float* wn_ = w.w0n + 16 * TT_ + 4 * q;                                                   \
            pw0[0] = *reinterpret_cast<const float4*>(wn_ + (size_t)(pdg.x < D ? pdg.x : 0) * Hp);         \
            pw0[1] = *reinterpret_cast<const float4*>(wn_ + (size_t)(pdg.y < D ? pdg.y : 0) * Hp);         \
            pw0[2] = *reinterpret_cast<const float4*>(wn_ + (size_t)(pdg.z < D ? pdg.z : 0) * Hp);         \
            pw0[3] = *reinterpret_cast<const float4*>(wn_ + (size_t)(pdg.w < D ? pdg.w : 0) * Hp);         \
            pb0 = *reinterpret_cast<const float4*>(w.b0 + 16 * TT_ + 4 * q);                               \
            pb1 = *reinterpret_cast<const float4*>(w.b1 + 16 * TT_ + 4 * q);                               \
            pb2 = *reinterpret_cast<const float4*>(w.b2 + 16 * TT_ + 4 * q);                               \
        }

        PREFETCH_N(0);

        // ---- rank 0 reads nothing: bias only
        {
            const f32x4 o0 = bias4(w.b3i, 4 * q), o1 = bias4(w.b3i, 16 + 4 * q);
            SOLVE_RANK(0, o0, o1)
        }

        for (int Tt = 0; Tt < nT; ++Tt) {
            const int4 dg = pdg;
            if (dg.x >= D && dg.y >= D && dg.z >= D && dg.w >= D) break;       // padding tiles

            // ---- bursts against everything that is already final, from prefetched fragments
            f32x4 a0, a1, a2;
            a0[0] = pb0.x; a0[1] = pb0.y; a0[2] = pb0.z; a0[3] = pb0.w;
            a1[0] = pb1.x; a1[1] = pb1.y; a1[2] = pb1.z; a1[3] = pb1.w;
            a2[0] = pb2.x; a2[1] = pb2.y; a2[2] = pb2.z; a2[3] = pb2.w;
#pragma unroll
            for (int i = 0; i < NPX; ++i) {
                if (i < nXT) {
                    const float4 b = *reinterpret_cast<const float4*>(X + (i << 8) + (lane << 2));
                    a0 = MFMA(pf0[i].x, b.x, a0); a0 = MFMA(pf0[i].y, b.y, a0);
                    a0 = MFMA(pf0[i].z, b.z, a0); a0 = MFMA(pf0[i].w, b.w, a0);
                }
            }
            for (int Xt = NPX; Xt < nXT; ++Xt) a0 = tile_mac(a0, w.f0 + (size_t)Tt * nXT * 64, X, Xt, lane);
#pragma unroll
            for (int i = 0; i < NPK; ++i) {
                if (i < Tt) {
                    const float4 b1 = *reinterpret_cast<const float4*>(H0 + (i << 8) + (lane << 2));
                    const float4 b2 = *reinterpret_cast<const float4*>(H1 + (i << 8) + (lane << 2));
                    a1 = MFMA(pf1[i].x, b1.x, a1); a2 = MFMA(pf2[i].x, b2.x, a2);
                    a1 = MFMA(pf1[i].y, b1.y, a1); a2 = MFMA(pf2[i].y, b2.y, a2);
                    a1 = MFMA(pf1[i].z, b1.z, a1); a2 = MFMA(pf2[i].z, b2.z, a2);
                    a1 = MFMA(pf1[i].w, b1.w, a1); a2 = MFMA(pf2[i].w, b2.w, a2);
                }
            }
            for (int K = NPK; K < Tt; ++K) {
                a1 = tile_mac(a1, w.f1 + (size_t)Tt * nT * 64, H0, K, lane);
                a2 = tile_mac(a2, w.f2 + (size_t)Tt * nT * 64, H1, K, lane);
            }
            const float4 d1 = pd1, d2 = pd2;
            const float4 w0r0 = pw0[0], w0r1 = pw0[1], w0r2 = pw0[2], w0r3 = pw0[3];

            // ---- everything of this tile is in registers: fetch the next tile's while the chains run
            if (Tt + 1 < nT) PREFETCH_N(Tt + 1);

            // ---- the degree groups of this tile, one after the other
            int j0 = 0;
            while (j0 < 4) {
                const int g = sel4i(dg, j0);
                int j1 = j0;
                while (j1 + 1 < 4 && sel4i(dg, j1 + 1) == g) ++j1;
                if (g >= D) { j0 = j1 + 1; continue; }
                const bool mine = (q >= j0) && (q <= j1);
                const int hb = (Tt << 8) + (lane << 2);          // B-operand base of this tile

                // the weight fragments of rank g's two output tiles against the final h2 tiles 0..Tt:
                // in flight while the hidden chain of the group runs
                float4 po0[NPO], po1[NPO];
                const float4* fo_ = w.f3i + ((size_t)g * 2 * nT) * 64 + lane;
                if (!(ABL & 8)) {
#pragma unroll
                for (int i = 0; i < NPO; ++i)
                    if (i <= Tt) { po0[i] = fo_[i * 64]; po1[i] = fo_[(nT + i) * 64]; }
                }
                f32x4 o0 = bias4(w.b3i, 32 * g + 4 * q), o1 = bias4(w.b3i, 32 * g + 16 + 4 * q);

                f32x4 h0, h1, h2;
                if (!(ABL & 4)) {
                for (int r = 0; r < 4; ++r) h0[r] = fmaxf(a0[r], 0.0f);
                if (mine) store_rows(H0, Tt, q, p, h0);
                WAVE_LDS_FENCE();
                DIAG4N(a1, d1, H0);
                for (int r = 0; r < 4; ++r) h1[r] = fmaxf(a1[r] + h0[r], 0.0f);
                if (mine) store_rows(H1, Tt, q, p, h1);
                WAVE_LDS_FENCE();
                DIAG4N(a2, d2, H1);
                for (int r = 0; r < 4; ++r) h2[r] = fmaxf(a2[r] + h1[r], 0.0f);
                if (mine) store_rows(H2, Tt, q, p, h2);
                WAVE_LDS_FENCE();
                }

                // ---- the 23 spline parameters of rank g: left-looking over the final h2 tiles
                if (!(ABL & 2)) {
                // (dealing the K < Tt part of this product into the hops of the chain above was measured: the
                // in-order queue then delays the chain's own dependent MFMAs -- 6 us slower)
#pragma unroll
                for (int i = 0; i < NPO; ++i) {
                    if (i <= Tt) {
                        const float4 b = *reinterpret_cast<const float4*>(H2 + (i << 8) + (lane << 2));
                        o0 = MFMA(po0[i].x, b.x, o0); o1 = MFMA(po1[i].x, b.x, o1);
                        o0 = MFMA(po0[i].y, b.y, o0); o1 = MFMA(po1[i].y, b.y, o1);
                        o0 = MFMA(po0[i].z, b.z, o0); o1 = MFMA(po1[i].z, b.z, o1);
                        o0 = MFMA(po0[i].w, b.w, o0); o1 = MFMA(po1[i].w, b.w, o1);
                    }
                }
                for (int K = NPO; K <= Tt; ++K) {
                    o0 = tile_mac(o0, w.f3i + ((size_t)g * 2 * nT) * 64, H2, K, lane);
                    o1 = tile_mac(o1, w.f3i + ((size_t)g * 2 * nT + nT) * 64, H2, K, lane);
                }
                }
                SOLVE_RANK(g, o0, o1)
                const float xg = X[lidx(g, p)];
                // ---- rank-1 update of this tile's layer-0 pre-activations
#define RANK1N(WV) { a0[0] += WV.x * xg; a0[1] += WV.y * xg; a0[2] += WV.z * xg; a0[3] += WV.w * xg; }
                if (j0 == 0) RANK1N(w0r0) else if (j0 == 1) RANK1N(w0r1) else if (j0 == 2) RANK1N(w0r2) else RANK1N(w0r3)
#undef RANK1N
                j0 = j1 + 1;
            }
        }
#undef PREFETCH_N
#undef SOLVE_RANK
        __syncthreads();
        const bool last = (t == 0);
        rerank_or_store(X, Y, out, row0, n, D, Dp, feat_of_rank + t * D,
                        last ? nullptr : rank_of_feat + (t - 1) * D, lane);
        __syncthreads();
    }
    if (ladj_out && lane < 16 && row0 + p < n) ladj_out[row0 + p] = ladj;
}

int pmc_launch_inverse_tri_nsf(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n,
                               hipStream_t stream) {
    const size_t lds = (size_t)(2 * m->Dp * 16 + 3 * m->Hp * 16 + 16 * 32 + 16 * 24) * sizeof(float);
    if (lds > 160 * 1024) return pmc_fail("pmc_maf_inverse: flow too wide for one wave's LDS budget (160 KiB)");
    static size_t lds_set = 0;
    if (lds > 48 * 1024 && lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(maf_inverse_tri_nsf_kernel<NSF_ABL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pmc_fail_hip(e, "hipFuncSetAttribute(maf_inverse_tri_nsf_kernel)");
        lds_set = lds;
    }
    hipLaunchKernelGGL(maf_inverse_tri_nsf_kernel<NSF_ABL>, dim3((unsigned)((n + 15) / 16)), dim3(64), lds, stream, *m, z, x,
                       ladj, n);
    return pmc_check_launch("maf_inverse_tri_nsf_kernel");
}

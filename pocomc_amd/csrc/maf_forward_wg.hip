// Flow.forward / Flow.log_prob (pocomc/flow.py:99-114, :134-147): data -> latent, for the affine
// (MAF) and the rational-quadratic-spline (NSF) univariate maps, and the reference's D-pass
// inverse for the spline flows (zuko AutoregressiveTransform._inverse: D fixed-point passes of the
// full hyper-network + one for the log-determinant).
//
// A workgroup of NW wavefronts owns 16 rows; the tiles of every layer are dealt to the waves
// (maf_wg.h), so a row set's latency is a fraction of the lone-wave kernel's and several waves per
// SIMD cover each other's weight-fetch latency.  NW = 8 for small batches (validation batches of
// Flow.fit, evidence draws), NW = 4 when there are enough row sets to fill the chip anyway.
//
// Spline flows: the output layer has 3 K - 1 rows per feature (K bins: 23 for the reference's 8).  It is
// produced 16 ranks (= exactly 3 K - 1 output tiles) at a time into an LDS panel, then one thread per
// (rank, row) evaluates its spline.  UNI: 0 = affine, else the number of bins (4, 8, 16).
#include <stdlib.h>
#include "maf_wg.h"
#include "rqs.h"

#define PANEL_TILES(UNI) RQS_NOUT_OF(UNI)   // 16 ranks x (3 K - 1) outputs = 3 K - 1 tiles of 16 rows

// out-layer panel of ranks [16c, 16c+16): tiles NOUT c .. NOUT c + NOUT - 1 -> P (LDS, local tile index)
template <int NW, int NOUT>
__device__ __forceinline__ void rqs_panel(const pmc_maf_t& m, const MafView& w, const float* H2, float* P, int c,
                                          int wv, int lane) {
    const int q = lane >> 4, p = lane & 15;
    for (int i = wv; i < NOUT; i += NW) {
        const int O = NOUT * c + i;
        if (16 * O >= NOUT * m.D) continue;                  // padding rows: never read
        f32x4 o = bias4(w.b3, 16 * O + 4 * q);
        o = mac_range<4>(o, w.f3 + (size_t)O * m.nT * 64, H2, 0, m.nT, lane);
        store_rows(P, i, q, p, o);
    }
}

// MODE 0: forward (x -> z, ladj of the forward map).  MODE 1: D-pass inverse (z -> x, ladj of the inverse map).
template <int NW, int UNI, int MODE>
__global__ __launch_bounds__(64 * NW) void maf_forward_wg_kernel(pmc_maf_t m, const float* __restrict__ in,
                                                                 float* __restrict__ out,
                                                                 float* __restrict__ ladj_out,
                                                                 float* __restrict__ logprob_out, int64_t n,
                                                                 const int64_t* __restrict__ idx) {
    constexpr bool PROF = false;
    constexpr int NOUT = UNI ? RQS_NOUT_OF(UNI) : 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, p = lane & 15;
    const int64_t row0 = (int64_t)blockIdx.x * 16;
    const int D = m.D, Dp = m.Dp, Hp = m.Hp, T = m.T, nT = m.nT, nOT = m.nOT, nXT = m.nXT;
    const int nOeff = min(nOT, (D + 7) / 8);
    float* Xc = smem;
    float* Xn = Xc + Dp * 16;
    float* A = Xn + Dp * 16;
    float* B = A + Hp * 16;
    float* C = B + Hp * 16;
    float* RED = C + Hp * 16;                 // [16 * NW]
    float* P = RED + 16 * NW;                 // spline: [NOUT * 256]
    float* Y = P + (UNI ? NOUT * 256 : 0);    // MODE 1: the transform's input, by rank
    const int* feat_of_rank = m.meta + 8;
    const int* rank_of_feat = m.meta + 8 + T * D;
    long long* pacc = nullptr; long long tk = 0;

    float* ld = MODE ? Y : Xc;
    const int* ford = MODE ? feat_of_rank + (T - 1) * D : feat_of_rank;
    for (int e = tid; e < Dp * 16; e += 64 * NW) {
        const int r = e >> 4, pp = e & 15;
        float v = 0.0f;
        if (r < D && row0 + pp < n) v = in[(idx ? idx[row0 + pp] : row0 + pp) * D + ford[r]];   // idx: row gather
        ld[lidx(r, pp)] = v;
        if (MODE) Xc[lidx(r, pp)] = 0.0f;
    }
    lds_barrier();
    float ladj = 0.0f;
    for (int tt = 0; tt < T; ++tt) {
        const int t = MODE ? T - 1 - tt : tt;
        const MafView w = maf_view(m, t);
        const bool last = (tt + 1 == T);
        const int npass = MODE ? D + 1 : 1;
        for (int pass = 0; pass < npass; ++pass) {
            const bool fin = (pass + 1 == npass);          // the pass whose log-determinant counts
            hidden_pass_wg<NW, 4, PROF>(m, w, Xc, A, B, C, wv, lane, pacc, tk);
            if (UNI == 0) {
                for (int O = wv; O < nOeff; O += NW) {
                    f32x4 o = bias4(w.b3, 16 * O + 4 * q);
                    o = mac_range<4>(o, w.f3 + (size_t)O * nT * 64, C, 0, nT, lane);
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int rank = 8 * O + 2 * q + s;
                        if (rank < D) {
                            const float shift = s ? o[2] : o[0];
                            const float ls = soft_ls(s ? o[3] : o[1]);
                            float y;
                            if (MODE) y = (Y[lidx(rank, p)] - shift) / expf(ls);
                            else y = Xc[lidx(rank, p)] * expf(ls) + shift;
                            const int feat = feat_of_rank[t * D + rank];
                            if (MODE && !fin) {
                                Xn[lidx(rank, p)] = y;
                            } else if (last) {
                                Xn[lidx(rank, p)] = y;
                                if (out && row0 + p < n) out[(row0 + p) * D + feat] = y;
                            } else {
                                // the next transform's rank order
                                Xn[lidx(rank_of_feat[(MODE ? t - 1 : t + 1) * D + feat], p)] = y;
                            }
                            if (fin) ladj += MODE ? -ls : ls;
                        }
                    }
                }
            } else {
                for (int c = 0; c < nXT; ++c) {
                    rqs_panel<NW, NOUT>(m, w, C, P, c, wv, lane);
                    lds_barrier();
                    for (int e = tid; e < 256; e += 64 * NW) {
                        const int rr = e >> 4, pp = e & 15, rank = 16 * c + rr;
                        if (rank < D) {
                            float phi[NOUT];
#pragma unroll
                            for (int j = 0; j < NOUT; ++j) phi[j] = P[lidx(NOUT * rr + j, pp)];
                            float y, l;
                            if (MODE) rqs_inverse_t<(UNI ? UNI : 8)>(phi, Y[lidx(rank, pp)], y, l);
                            else rqs_forward_t<(UNI ? UNI : 8)>(phi, Xc[lidx(rank, pp)], y, l);
                            const int feat = feat_of_rank[t * D + rank];
                            if (MODE && !fin) {
                                Xn[lidx(rank, pp)] = y;
                            } else if (last) {
                                Xn[lidx(rank, pp)] = y;
                                if (out && row0 + pp < n) out[(row0 + pp) * D + feat] = y;
                            } else {
                                Xn[lidx(rank_of_feat[(MODE ? t - 1 : t + 1) * D + feat], pp)] = y;
                            }
                            if (fin) ladj += MODE ? -l : l;
                        }
                    }
                    lds_barrier();
                }
            }
            for (int e = tid; e < (Dp - D) * 16; e += 64 * NW) Xn[lidx(D + (e >> 4), e & 15)] = 0.0f;
            lds_barrier();
            if (MODE && fin && !last) {
                // Xn holds x of this transform in the next transform's rank order: it becomes that
                // transform's Y; its iterate starts from zero
                for (int e = tid; e < Dp * 16; e += 64 * NW) { Y[e] = Xn[e]; Xn[e] = 0.0f; }
                lds_barrier();
            }
            float* sw = Xc; Xc = Xn; Xn = sw;
        }
    }
    const float l = quad_sum(ladj);
    if (lane < 16) RED[wv * 16 + lane] = l;
    lds_barrier();
    if (wv == 0) {
        float lt = 0.0f;
#pragma unroll
        for (int k = 0; k < NW; ++k) lt += RED[16 * k + p];
        if (ladj_out && lane < 16 && row0 + p < n) ladj_out[row0 + p] = lt;
        if (MODE == 0 && logprob_out) {
            // base N(0,I) log-density of z (flow.py:147 -> zuko DiagNormal)
            float ss = 0.0f;
            for (int r = q; r < D; r += 4) { const float z = Xc[lidx(r, p)]; ss += z * z; }
            ss = quad_sum(ss);
            if (lane < 16 && row0 + p < n)
                logprob_out[row0 + p] = (-0.5f * ss - 0.9189385332046727f * (float)D) + lt;
        }
    }
}

template <int NW, int UNI, int MODE>
static int launch_forward_wg(const pmc_maf_t* m, const float* x, float* z, float* ladj, float* log_prob, int64_t n,
                             hipStream_t st, const int64_t* idx = nullptr) {
    const size_t lds = (size_t)(2 * m->Dp * 16 + 3 * m->Hp * 16 + 16 * NW + (UNI ? PANEL_TILES(UNI) * 256 : 0) +
                                (MODE ? m->Dp * 16 : 0)) * sizeof(float);
    if (lds > 160 * 1024) return pmc_fail("pmc_maf_forward: flow too wide for 160 KB of LDS");
    static size_t lds_set = 0;
    if (lds > 48 * 1024 && lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(maf_forward_wg_kernel<NW, UNI, MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pmc_fail_hip(e, "hipFuncSetAttribute(maf_forward_wg_kernel)");
        lds_set = lds;
    }
    hipLaunchKernelGGL((maf_forward_wg_kernel<NW, UNI, MODE>), dim3((unsigned)((n + 15) / 16)), dim3(64 * NW), lds, st,
                       *m, x, z, ladj, log_prob, n, idx);
    return pmc_check_launch("maf_forward_wg_kernel");
}

int pmc_launch_forward_wg(const pmc_maf_t* m, const float* x, float* z, float* ladj, float* log_prob, int64_t n,
                          hipStream_t st, const int64_t* idx) {
    // enough row sets to give every SIMD a few waves anyway -> fewer waves per set (less barrier idling)
    static const int force = pmc_env_int("PMC_FWD_NW", 0);      // A/B switch
    const bool wide = force ? force == 8 : n <= 16 * 1024;
#define FWD_SPLINE(K)                                                                                   \
    if (m->n_out == RQS_NOUT_OF(K))                                                                      \
        return wide ? launch_forward_wg<8, K, 0>(m, x, z, ladj, log_prob, n, st, idx)                    \
                    : launch_forward_wg<4, K, 0>(m, x, z, ladj, log_prob, n, st, idx);
    FWD_SPLINE(8) FWD_SPLINE(4) FWD_SPLINE(16)
#undef FWD_SPLINE
    return wide ? launch_forward_wg<8, 0, 0>(m, x, z, ladj, log_prob, n, st, idx)
                : launch_forward_wg<4, 0, 0>(m, x, z, ladj, log_prob, n, st, idx);
}

// D-pass inverse through the same workgroup kernel (the only inverse of the spline flows until the
// triangular sweep learns the spline; a cross-check for the affine flows)
int pmc_launch_inverse_dpass_wg(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n, hipStream_t st) {
    if (m->n_out == RQS_NOUT_OF(8)) return launch_forward_wg<8, 8, 1>(m, z, x, ladj, nullptr, n, st);
    if (m->n_out == RQS_NOUT_OF(4)) return launch_forward_wg<8, 4, 1>(m, z, x, ladj, nullptr, n, st);
    if (m->n_out == RQS_NOUT_OF(16)) return launch_forward_wg<8, 16, 1>(m, z, x, ladj, nullptr, n, st);
    return launch_forward_wg<8, 0, 1>(m, z, x, ladj, nullptr, n, st);
}

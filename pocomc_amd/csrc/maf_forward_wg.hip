// Flow.forward / Flow.log_prob (pocomc/flow.py:99-114, :134-147): data -> latent.
//
// A workgroup of NW wavefronts owns 16 rows; the tiles of every layer are dealt to the waves
// (maf_wg.h), so a row set's latency is a fraction of the lone-wave kernel's and several waves per
// SIMD cover each other's weight-fetch latency.  NW = 8 for small batches (validation batches of
// Flow.fit, evidence draws), NW = 4 when there are enough row sets to fill the chip anyway.
#include <stdlib.h>
#include "maf_wg.h"

template <int NW>
__global__ __launch_bounds__(64 * NW) void maf_forward_wg_kernel(pmc_maf_t m, const float* __restrict__ in,
                                                                 float* __restrict__ out,
                                                                 float* __restrict__ ladj_out,
                                                                 float* __restrict__ logprob_out, int64_t n) {
    constexpr bool PROF = false;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, p = lane & 15;
    const int64_t row0 = (int64_t)blockIdx.x * 16;
    const int D = m.D, Dp = m.Dp, Hp = m.Hp, T = m.T, nT = m.nT, nOT = m.nOT;
    const int nOeff = min(nOT, (D + 7) / 8);
    float* Xc = smem;
    float* Xn = Xc + Dp * 16;
    float* A = Xn + Dp * 16;
    float* B = A + Hp * 16;
    float* C = B + Hp * 16;
    float* RED = C + Hp * 16;                 // [16 * NW]
    const int* feat_of_rank = m.meta + 8;
    const int* rank_of_feat = m.meta + 8 + T * D;
    long long* pacc = nullptr; long long tk = 0;

    for (int e = tid; e < Dp * 16; e += 64 * NW) {
        const int r = e >> 4, pp = e & 15;
        float v = 0.0f;
        if (r < D && row0 + pp < n) v = in[(row0 + pp) * D + feat_of_rank[r]];
        Xc[lidx(r, pp)] = v;
    }
    lds_barrier();
    float ladj = 0.0f;
    for (int t = 0; t < T; ++t) {
        const MafView w = maf_view(m, t);
        const bool last = (t + 1 == T);
        hidden_pass_wg<NW, 4, PROF>(m, w, Xc, A, B, C, wv, lane, pacc, tk);
        for (int O = wv; O < nOeff; O += NW) {
            f32x4 o = bias4(w.b3, 16 * O + 4 * q);
            o = mac_range<4>(o, w.f3 + (size_t)O * nT * 64, C, 0, nT, lane);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int rank = 8 * O + 2 * q + s;
                if (rank < D) {
                    const float shift = s ? o[2] : o[0];
                    const float ls = soft_ls(s ? o[3] : o[1]);
                    const float y = Xc[lidx(rank, p)] * expf(ls) + shift;
                    const int feat = feat_of_rank[t * D + rank];
                    if (last) {
                        Xn[lidx(rank, p)] = y;
                        if (row0 + p < n) out[(row0 + p) * D + feat] = y;
                    } else {
                        Xn[lidx(rank_of_feat[(t + 1) * D + feat], p)] = y;   // the next transform's rank order
                    }
                    ladj += ls;
                }
            }
        }
        for (int e = tid; e < (Dp - D) * 16; e += 64 * NW) Xn[lidx(D + (e >> 4), e & 15)] = 0.0f;
        lds_barrier();
        float* sw = Xc; Xc = Xn; Xn = sw;
    }
    const float l = quad_sum(ladj);
    if (lane < 16) RED[wv * 16 + lane] = l;
    lds_barrier();
    if (wv == 0) {
        float lt = 0.0f;
#pragma unroll
        for (int k = 0; k < NW; ++k) lt += RED[16 * k + p];
        if (ladj_out && lane < 16 && row0 + p < n) ladj_out[row0 + p] = lt;
        if (logprob_out) {
            // base N(0,I) log-density of z (flow.py:147 -> zuko DiagNormal)
            float ss = 0.0f;
            for (int r = q; r < D; r += 4) { const float z = Xc[lidx(r, p)]; ss += z * z; }
            ss = quad_sum(ss);
            if (lane < 16 && row0 + p < n)
                logprob_out[row0 + p] = (-0.5f * ss - 0.9189385332046727f * (float)D) + lt;
        }
    }
}

template <int NW>
static int launch_forward_wg(const pmc_maf_t* m, const float* x, float* z, float* ladj, float* log_prob, int64_t n,
                             hipStream_t st) {
    const size_t lds = (size_t)(2 * m->Dp * 16 + 3 * m->Hp * 16 + 16 * NW) * sizeof(float);
    if (lds > 160 * 1024) return pmc_fail("pmc_maf_forward: flow too wide for 160 KB of LDS");
    static size_t lds_set = 0;
    if (lds > 48 * 1024 && lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(maf_forward_wg_kernel<NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pmc_fail_hip(e, "hipFuncSetAttribute(maf_forward_wg_kernel)");
        lds_set = lds;
    }
    hipLaunchKernelGGL(maf_forward_wg_kernel<NW>, dim3((unsigned)((n + 15) / 16)), dim3(64 * NW), lds, st, *m, x, z,
                       ladj, log_prob, n);
    return pmc_check_launch("maf_forward_wg_kernel");
}

int pmc_launch_forward_wg(const pmc_maf_t* m, const float* x, float* z, float* ladj, float* log_prob, int64_t n,
                          hipStream_t st) {
    // enough row sets to give every SIMD a few waves anyway -> fewer waves per set (less barrier idling)
    static const int force = getenv("PMC_FWD_NW") ? atoi(getenv("PMC_FWD_NW")) : 0;      // A/B switch
    if (force == 2) return launch_forward_wg<2>(m, x, z, ladj, log_prob, n, st);
    if (force == 4) return launch_forward_wg<4>(m, x, z, ladj, log_prob, n, st);
    if (force == 8) return launch_forward_wg<8>(m, x, z, ladj, log_prob, n, st);
    if (n > 16 * 1024) return launch_forward_wg<4>(m, x, z, ladj, log_prob, n, st);
    return launch_forward_wg<8>(m, x, z, ladj, log_prob, n, st);
}

// Pieces shared by the kernels in which a WORKGROUP of NW wavefronts owns 16 rows (flow forward,
// flow training): the tiles of a layer are dealt to the waves, activations sit in workgroup LDS in
// the MFMA operand layout, an LDS-only barrier separates dependent layers.
#ifndef PMC_MAF_WG_H
#define PMC_MAF_WG_H

#include "maf_common.h"

// Workgroup barrier that orders LDS traffic only: global stores (activation / delta scratch) and prefetched
// weight loads stay in flight across it (a __syncthreads() would drain vmcnt to zero).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Deal tiles whose cost falls with `r` to the waves in a snake, so the sums balance.
template <int NW>
__device__ __forceinline__ int snake_owner(int r) {
    const int pos = r % NW;
    return ((r / NW) & 1) ? NW - 1 - pos : pos;
}
// the i-th item (cost rank) wave `wv` owns under snake_owner: wv, 2NW-1-wv, 2NW+wv, 4NW-1-wv, ... (increasing)
template <int NW>
__device__ __forceinline__ int snake_item(int wv, int i) {
    return (i & 1) ? (i + 1) * NW - 1 - wv : i * NW + wv;
}

__device__ __forceinline__ f32x4 rows_of(const float* H, int T, int q, int p) {
    const float* hb = H + (T << 8) + (p << 2) + q;
    f32x4 a = {hb[0], hb[64], hb[128], hb[192]};
    return a;
}

__device__ __forceinline__ f32x4 relu_gate(f32x4 a, const float* H, int T, int q, int p) {
    const f32x4 h = rows_of(H, T, q, p);
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = h[r] > 0.f ? a[r] : 0.f;
    return a;
}

#define TICKT() (PROF ? (long long)__builtin_readcyclecounter() : 0LL)
#define LAPT(I) if (PROF) { const long long t2_ = TICKT(); pacc[I] += t2_ - tk; tk = t2_; }
// phase barrier: time up to the barrier goes to phase I, the wait itself to slot 10
#define PHASE_END(I) { LAPT(I) lds_barrier(); LAPT(10) }

// hidden layers of one transform for the workgroup's 16 rows: X -> H0, H1, H2 (tiles dealt to the waves)
template <int NW, int PF, bool PROF>
__device__ __forceinline__ void hidden_pass_wg(const pmc_maf_t& m, const MafView& w, const float* X, float* H0,
                                               float* H1, float* H2, int wv, int lane, long long* pacc,
                                               long long& tk) {
    const int q = lane >> 4, p = lane & 15;
    const int nT = m.nT, nXT = m.nXT;
    for (int T = wv; T < nT; T += NW) {
        f32x4 a = bias4(w.b0, 16 * T + 4 * q);
        a = mac_range<PF>(a, w.f0 + (size_t)T * nXT * 64, X, 0, nXT, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = fmaxf(a[r], 0.0f);
        store_rows(H0, T, q, p, a);
    }
    LAPT(15)
    lds_barrier();
    LAPT(15)
    for (int layer = 1; layer <= 2; ++layer) {
        const float* Hin = layer == 1 ? H0 : H1;
        float* Hout = layer == 1 ? H1 : H2;
        const float4* f = layer == 1 ? w.f1 : w.f2;
        const float* b = layer == 1 ? w.b1 : w.b2;
        for (int it = 0;; ++it) {                                  // the tiles this wave owns, most expensive first
            const int r = snake_item<NW>(wv, it);
            if (r >= nT) break;
            const int T = nT - 1 - r;
            f32x4 a = bias4(b, 16 * T + 4 * q);
            if constexpr (PROF) {
                // timing build only: fragments first, explicit waits, so that load latency (slot 11), the MFMA
                // chain (12) and the epilogue (13) show up separately
                const int kend = m.tri_ok ? T + 1 : nT;
                const float4* ff = f + (size_t)T * nT * 64 + lane;
                float4 fr[12];
                LAPT(15)
#pragma unroll
                for (int j = 0; j < 12; ++j) if (j < kend) fr[j] = ff[j * 64];
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                LAPT(11)
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    if (j < kend) {
                        const float4 b_ = *reinterpret_cast<const float4*>(Hin + (j << 8) + (lane << 2));
                        a = MFMA(fr[j].x, b_.x, a); a = MFMA(fr[j].y, b_.y, a);
                        a = MFMA(fr[j].z, b_.z, a); a = MFMA(fr[j].w, b_.w, a);
                    }
                }
                for (int K = 12; K < kend; ++K) a = tile_mac(a, f + (size_t)T * nT * 64, Hin, K, lane);
                asm volatile("s_nop 0" :: "v"(a[0]));
                LAPT(12)
            } else {
                a = mac_range<PF>(a, f + (size_t)T * nT * 64, Hin, 0, m.tri_ok ? T + 1 : nT, lane);
            }
            const f32x4 h = rows_of(Hin, T, q, p);
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = fmaxf(a[r] + h[r], 0.0f);
            store_rows(Hout, T, q, p, a);
        }
        LAPT(13)
        lds_barrier();
        LAPT(14)
    }
}


#endif

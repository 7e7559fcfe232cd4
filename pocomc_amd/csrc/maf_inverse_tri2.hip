// Triangular-sweep MAF inverse, latency-tuned version (the kernel of every MCMC step,
// pocomc/mcmc.py:88 -> flow.py:116-132).
//
// Same algorithm and data layout as maf_inverse_tri_kernel (maf_kernels.hip); what changes is
// how latency is hidden, because at the headline size (1e4 walkers = 625 waves on 1024 SIMDs)
// every wave is alone on its SIMD and nothing else covers a stall:
//   * one wave per workgroup and NO barriers inside the sweep: a wave's DS instructions run in
//     issue order, so dependent ds_write -> ds_read pairs only need the compiler to keep program
//     order (WAVE_LDS_FENCE); an s_barrier's implicit vmcnt(0) would drain the prefetches below;
//   * the weight fragments, biases, diagonal blocks and W0 rows a tile needs are fetched into
//     registers while the PREVIOUS tile's dependent chain runs (one wave may use all 512 VGPRs),
//     so the L2 latency of ~30 KiB of weights per tile never sits on the critical path;
//   * the per-group chain  h0 -> h1 -> h2 -> (shift, raw) -> x  reads no global memory at all.

#include "maf_common.h"

// acc += W[tile][quad j] * act[quad j] for the quads j0..j1 of the current group; j is a
// compile-time constant in every arm (a run-time index into a float4 goes through scratch)
#define DIAG4(ACC, FRAG, ACT)                                             \
    {                                                                     \
        if (j0 <= 0 && 0 <= j1) ACC = MFMA(FRAG.x, ACT[hb + 0], ACC);     \
        if (j0 <= 1 && 1 <= j1) ACC = MFMA(FRAG.y, ACT[hb + 1], ACC);     \
        if (j0 <= 2 && 2 <= j1) ACC = MFMA(FRAG.z, ACT[hb + 2], ACC);     \
        if (j0 <= 3 && 3 <= j1) ACC = MFMA(FRAG.w, ACT[hb + 3], ACC);     \
    }

#define PX 2     // x tiles (16 ranks each) prefetched per hidden tile   -> D  <= 32 fully prefetched
#define PK 8     // K tiles prefetched per hidden tile and layer          -> Hp <= 144 fully prefetched

// PROF: accumulate s_memtime cycles per phase into prof[wave][8] (debug builds of the bench only)
#define TICK() (PROF ? (long long)__builtin_readcyclecounter() : 0LL)

template <bool PROF>
__global__ __launch_bounds__(64) void maf_inverse_tri2_kernel(pmc_maf_t m, const float* __restrict__ in,
                                                              float* __restrict__ out,
                                                              float* __restrict__ ladj_out, int64_t n,
                                                              long long* __restrict__ prof) {
    long long c_zero = 0, c_burst = 0, c_pref = 0, c_chain = 0, c_oburst = 0, c_x = 0, c_tail = 0;
    const long long c_begin = TICK();
    const long long w_begin = PROF ? (long long)wall_clock64() : 0LL;
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
    const int* feat_of_rank = m.meta + 8;
    const int* rank_of_feat = m.meta + 8 + T * D;
    const int* quad_meta = m.meta + 8 + 2 * T * D;

    load_rows(Y, in, row0, n, D, Dp, feat_of_rank + (T - 1) * D, lane);
    float ladj = 0.0f;               // per-lane partial: owner lanes add their ranks

    for (int t = T - 1; t >= 0; --t) {
        const MafView w = maf_view(m, t);
        long long tk = TICK();
        {
            float4* z4 = reinterpret_cast<float4*>(X);
            const int n4 = (Dp * 16 + 3 * Hp * 16) >> 2;
            for (int e = lane; e < n4; e += 64) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();

        // ---------------- prefetch registers (filled for tile Tt while tile Tt-1 runs)
        float4 pf0[PX], pf1[PK], pf2[PK], pw0[4];
        float4 pd1, pd2, pd3, pb0, pb1, pb2;
        int4 pdg;
        int po = 0;                                   // output tile pd3 was fetched for

#define PREFETCH(TT)                                                                                       \
        {                                                                                                  \
            const int TT_ = (TT);                                                                          \
            pdg = *reinterpret_cast<const int4*>(quad_meta + 4 * TT_);                                     \
            pdg.x &= 0xffff; pdg.y &= 0xffff; pdg.z &= 0xffff; pdg.w &= 0xffff;                            \
            const float4* f0_ = w.f0 + ((size_t)TT_ * nXT) * 64 + lane;                                    \
            _Pragma("unroll") for (int i_ = 0; i_ < PX; ++i_) if (i_ < nXT) pf0[i_] = f0_[i_ * 64];        \
            const float4* f1_ = w.f1 + ((size_t)TT_ * nT) * 64 + lane;                                     \
            const float4* f2_ = w.f2 + ((size_t)TT_ * nT) * 64 + lane;                                     \
            _Pragma("unroll") for (int i_ = 0; i_ < PK; ++i_) if (i_ < TT_) { pf1[i_] = f1_[i_ * 64]; pf2[i_] = f2_[i_ * 64]; } \
            pd1 = f1_[TT_ * 64]; pd2 = f2_[TT_ * 64];                                                      \
            const int g0_ = pdg.x < D ? pdg.x : (pdg.y < D ? pdg.y : (pdg.z < D ? pdg.z : pdg.w));         \
            po = (g0_ < D ? g0_ : 0) >> 3;                                                                 \
            pd3 = w.f3[((size_t)po * nT + TT_) * 64 + lane];                                               \
            const float* wn_ = w.w0n + 16 * TT_ + 4 * q;                                                   \
            pw0[0] = *reinterpret_cast<const float4*>(wn_ + (size_t)(pdg.x < D ? pdg.x : 0) * Hp);         \
            pw0[1] = *reinterpret_cast<const float4*>(wn_ + (size_t)(pdg.y < D ? pdg.y : 0) * Hp);         \
            pw0[2] = *reinterpret_cast<const float4*>(wn_ + (size_t)(pdg.z < D ? pdg.z : 0) * Hp);         \
            pw0[3] = *reinterpret_cast<const float4*>(wn_ + (size_t)(pdg.w < D ? pdg.w : 0) * Hp);         \
            pb0 = *reinterpret_cast<const float4*>(w.b0 + 16 * TT_ + 4 * q);                               \
            pb1 = *reinterpret_cast<const float4*>(w.b1 + 16 * TT_ + 4 * q);                               \
            pb2 = *reinterpret_cast<const float4*>(w.b2 + 16 * TT_ + 4 * q);                               \
        }

        PREFETCH(0);

        int o_cur = 0;
        f32x4 oacc = bias4(w.b3, 4 * q);
        // ---- rank 0 reads nothing: bias only
        {
            const float yv = Y[lidx(0, p)];
            const float ls = soft_ls(oacc[1]);
            const float xv = (yv - oacc[0]) / expf(ls);
            if (q == 0) { X[lidx(0, p)] = xv; ladj -= ls; }
        }
        WAVE_LDS_FENCE();
        if (PROF) { const long long t2 = TICK(); c_zero += t2 - tk; tk = t2; }

        for (int Tt = 0; Tt < nT; ++Tt) {
            const int4 dg = pdg;
            if (dg.x >= D && dg.y >= D && dg.z >= D && dg.w >= D) break;       // padding tiles

            // ---- bursts against everything that is already final, from prefetched fragments
            f32x4 a0, a1, a2;
            a0[0] = pb0.x; a0[1] = pb0.y; a0[2] = pb0.z; a0[3] = pb0.w;
            a1[0] = pb1.x; a1[1] = pb1.y; a1[2] = pb1.z; a1[3] = pb1.w;
            a2[0] = pb2.x; a2[1] = pb2.y; a2[2] = pb2.z; a2[3] = pb2.w;
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                if (i < nXT) {
                    const float4 b = *reinterpret_cast<const float4*>(X + (i << 8) + (lane << 2));
                    a0 = MFMA(pf0[i].x, b.x, a0); a0 = MFMA(pf0[i].y, b.y, a0);
                    a0 = MFMA(pf0[i].z, b.z, a0); a0 = MFMA(pf0[i].w, b.w, a0);
                }
            }
            for (int Xt = PX; Xt < nXT; ++Xt) a0 = tile_mac(a0, w.f0 + (size_t)Tt * nXT * 64, X, Xt, lane);
#pragma unroll
            for (int i = 0; i < PK; ++i) {
                if (i < Tt) {
                    const float4 b1 = *reinterpret_cast<const float4*>(H0 + (i << 8) + (lane << 2));
                    const float4 b2 = *reinterpret_cast<const float4*>(H1 + (i << 8) + (lane << 2));
                    a1 = MFMA(pf1[i].x, b1.x, a1); a2 = MFMA(pf2[i].x, b2.x, a2);
                    a1 = MFMA(pf1[i].y, b1.y, a1); a2 = MFMA(pf2[i].y, b2.y, a2);
                    a1 = MFMA(pf1[i].z, b1.z, a1); a2 = MFMA(pf2[i].z, b2.z, a2);
                    a1 = MFMA(pf1[i].w, b1.w, a1); a2 = MFMA(pf2[i].w, b2.w, a2);
                }
            }
            for (int K = PK; K < Tt; ++K) {
                a1 = tile_mac(a1, w.f1 + (size_t)Tt * nT * 64, H0, K, lane);
                a2 = tile_mac(a2, w.f2 + (size_t)Tt * nT * 64, H1, K, lane);
            }
            const float4 d1 = pd1, d2 = pd2;
            float4 d3 = pd3;
            int d3_o = po;
            const float4 w0r0 = pw0[0], w0r1 = pw0[1], w0r2 = pw0[2], w0r3 = pw0[3];

            if (PROF) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long t2 = TICK(); c_burst += t2 - tk; tk = t2; }
            // ---- everything of this tile is in registers: fetch the next tile's while the chain runs
            if (Tt + 1 < nT) PREFETCH(Tt + 1);
            if (PROF) { const long long t2 = TICK(); c_pref += t2 - tk; tk = t2; }

            // ---- the degree groups of this tile, one after the other
            int j0 = 0;
            while (j0 < 4) {
                const int g = sel4i(dg, j0);
                int j1 = j0;
                while (j1 + 1 < 4 && sel4i(dg, j1 + 1) == g) ++j1;
                if (g >= D) { j0 = j1 + 1; continue; }
                const bool mine = (q >= j0) && (q <= j1);
                const int hb = (Tt << 8) + (lane << 2);          // B-operand base of this tile

                f32x4 h0, h1, h2;
                for (int r = 0; r < 4; ++r) h0[r] = fmaxf(a0[r], 0.0f);
                if (mine) store_rows(H0, Tt, q, p, h0);
                WAVE_LDS_FENCE();
                DIAG4(a1, d1, H0);
                for (int r = 0; r < 4; ++r) h1[r] = fmaxf(a1[r] + h0[r], 0.0f);
                if (mine) store_rows(H1, Tt, q, p, h1);
                WAVE_LDS_FENCE();
                DIAG4(a2, d2, H1);
                for (int r = 0; r < 4; ++r) h2[r] = fmaxf(a2[r] + h1[r], 0.0f);
                if (mine) store_rows(H2, Tt, q, p, h2);
                WAVE_LDS_FENCE();

                if (PROF) { const long long t2 = TICK(); c_chain += t2 - tk; tk = t2; }
                // ---- (shift, raw) of rank g
                if ((g & 7) == 0) {
                    o_cur = g >> 3;                               // new output tile: left-looking burst
                    oacc = bias4(w.b3, 16 * o_cur + 4 * q);
                    for (int K = 0; K <= Tt; ++K) oacc = tile_mac(oacc, w.f3 + (size_t)o_cur * nT * 64, H2, K, lane);
                } else {
                    if (d3_o != o_cur) {                          // tile straddles two output tiles
                        d3 = w.f3[((size_t)o_cur * nT + Tt) * 64 + lane];
                        d3_o = o_cur;
                    }
                    DIAG4(oacc, d3, H2);
                }
                if (PROF) { const long long t2 = TICK(); c_oburst += t2 - tk; tk = t2; }
                {
                    const int s = g & 1, qo = (g & 7) >> 1;
                    const float shift = s ? oacc[2] : oacc[0];
                    const float ls = soft_ls(s ? oacc[3] : oacc[1]);
                    const float yv = Y[lidx(g, p)];
                    const float xv = (yv - shift) / expf(ls);
                    if (q == qo) { X[lidx(g, p)] = xv; ladj -= ls; }
                }
                WAVE_LDS_FENCE();
                const float xg = X[lidx(g, p)];
                // ---- rank-1 update of this tile's layer-0 pre-activations
#define RANK1(WV) { a0[0] += WV.x * xg; a0[1] += WV.y * xg; a0[2] += WV.z * xg; a0[3] += WV.w * xg; }
                if (j0 == 0) RANK1(w0r0) else if (j0 == 1) RANK1(w0r1) else if (j0 == 2) RANK1(w0r2) else RANK1(w0r3)
#undef RANK1
                if (PROF) { const long long t2 = TICK(); c_x += t2 - tk; tk = t2; }
                j0 = j1 + 1;
            }
        }
#undef PREFETCH
        __syncthreads();
        const bool last = (t == 0);
        rerank_or_store(X, Y, out, row0, n, D, Dp, feat_of_rank + t * D,
                        last ? nullptr : rank_of_feat + (t - 1) * D, lane);
        __syncthreads();
    }
    ladj = quad_sum(ladj);
    if (ladj_out && lane < 16 && row0 + p < n) ladj_out[row0 + p] = ladj;
    if (PROF && lane == 0) {
        long long* P = prof + (size_t)blockIdx.x * 8;
        P[0] = TICK() - c_begin; P[1] = c_zero; P[2] = c_burst; P[3] = c_pref; P[4] = c_chain; P[5] = c_oburst;
        P[6] = c_x; P[7] = (long long)wall_clock64() - w_begin;
    }
    (void)c_tail;
}

extern "C" int pmc_debug_inverse_profile(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n,
                                         long long* prof, void* stream) {
    const size_t lds = (size_t)(2 * m->Dp * 16 + 3 * m->Hp * 16) * sizeof(float);
    hipLaunchKernelGGL(maf_inverse_tri2_kernel<true>, dim3((unsigned)((n + 15) / 16)), dim3(64), lds,
                       (hipStream_t)stream, *m, z, x, ladj, n, prof);
    return pmc_check_launch("maf_inverse_tri2_kernel<prof>");
}

int pmc_launch_inverse_tri2(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n, size_t lds,
                            hipStream_t stream) {
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(maf_inverse_tri2_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pmc_fail_hip(e, "hipFuncSetAttribute(maf_inverse_tri2_kernel)");
    }
    hipLaunchKernelGGL(maf_inverse_tri2_kernel<false>, dim3((unsigned)((n + 15) / 16)), dim3(64), lds, stream, *m, z,
                       x, ladj, n, (long long*)nullptr);
    return pmc_check_launch("maf_inverse_tri2_kernel");
}

// Triangular-sweep MAF inverse with a REGISTER-RESIDENT dependent chain.
//
// maf_inverse_tri2_kernel spends most of its time on LDS round trips: the MFMA C layout (lane
// (q,p) holds rows 4q..4q+3 of walker p) is not the B-operand layout (lane (k,p) supplies row k),
// so every layer hop of the per-degree-group chain goes  regs -> ds_write -> ds_read -> MFMA
// (~450 cycles per hop, measured with scripts/profile_inverse.py).
//
// Here the chain runs in an "R layout": the A operand of a chain MFMA carries the 4 output rows of
// ONE quad replicated over the 16 tile rows (A[i][k] = W[quad row (i&3)][k]), so after the MFMA
// every lane -- whatever its q -- holds all 4 values of that quad for its walker p.  Then
//   * the next hop's B operand (row k = q) is a register select, no LDS;
//   * residual adds, ReLU, (shift, raw), x_rank and the rank-1 update of layer 0 are lane-local;
//   * LDS is only WRITTEN (one ds_write_b32 per quad and layer) for the left-looking bursts of
//     later tiles, never read back on the chain.
// Contributions from previous tiles still come from natural-layout MFMA bursts (full 16-row
// efficiency) and are moved into the R layout once per tile through a 3 KiB LDS staging block.
// The output layer keeps natural-layout accumulators for all output tiles, updated right-looking
// by every finished quad (B operand = the same register select), and the current tile's own
// contributions in R-layout "pair slots" (two ranks per MFMA).
//
// Per tile the group structure (which quads form a degree group) is a compile-time pattern PAT:
// bit j set = quad j starts a new group; the chain is fully unrolled per pattern.

#include "maf_chain.h"

#define PX3 2
#define PK3 8

template <int MAXO>
struct NextFrags {
    float4 wd1[4], wd2[4], wo[2], f3n[MAXO], w0r[4][4];
    int4 dg;
    int g[4];
};

#define LOAD_CHAIN_FRAGS(NX, TT)                                                                              \
        {                                                                                                     \
            const int TT_ = (TT);                                                                             \
            int4 dg_ = *reinterpret_cast<const int4*>(quad_meta + 4 * TT_);                                   \
            dg_.x &= 0xffff; dg_.y &= 0xffff; dg_.z &= 0xffff; dg_.w &= 0xffff;                               \
            NX.dg = dg_;                                                                                      \
            const bool ny = dg_.y != dg_.x, nz = dg_.z != dg_.y, nw = dg_.w != dg_.z;                         \
            NX.g[0] = dg_.x;                                                                                  \
            NX.g[1] = ny ? dg_.y : (nz ? dg_.z : (nw ? dg_.w : D));                                           \
            NX.g[2] = ny ? (nz ? dg_.z : (nw ? dg_.w : D)) : ((nz && nw) ? dg_.w : D);                        \
            NX.g[3] = (ny && nz && nw) ? dg_.w : D;                                                           \
            _Pragma("unroll") for (int jt = 0; jt < 4; ++jt) {                                                \
                NX.wd1[jt] = w.f1[((size_t)TT_ * nT + TT_) * 64 + rl_hidden + 4 * jt];                        \
                NX.wd2[jt] = w.f2[((size_t)TT_ * nT + TT_) * 64 + rl_hidden + 4 * jt];                        \
            }                                                                                                 \
            _Pragma("unroll") for (int sl = 0; sl < 2; ++sl) {                                                \
                const int g_even = NX.g[2 * sl], g_odd = NX.g[2 * sl + 1];                                    \
                const int gsel = (lane & 2) ? g_odd : g_even;                                                 \
                const bool ok = gsel < D;                                                                     \
                const int gg = ok ? gsel : 0;                                                                 \
                const float4 v = w.f3[((size_t)(gg >> 3) * nT + TT_) * 64 + (q << 4) + 2 * (gg & 7) + (lane & 1)]; \
                NX.wo[sl] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);                                         \
            }                                                                                                 \
            _Pragma("unroll") for (int O = 0; O < MAXO; ++O)                                                  \
                NX.f3n[O] = (O < nOT) ? w.f3[((size_t)O * nT + TT_) * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f); \
            _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                   \
                const int gg = NX.g[i] < D ? NX.g[i] : 0;                                                     \
                _Pragma("unroll") for (int jt = i + 1; jt < 4; ++jt)                                          \
                    NX.w0r[i][jt] = *reinterpret_cast<const float4*>(w.w0n + (size_t)gg * Hp + 16 * TT_ + 4 * jt); \
            }                                                                                                 \
        }

#define TICK3() (PROF ? (long long)__builtin_readcyclecounter() : 0LL)
#define LAP(ACC) if (PROF) { const long long t2_ = TICK3(); ACC += t2_ - tk; tk = t2_; }

template <int MAXO, bool PROF>
__global__ __launch_bounds__(64) void maf_inverse_tri3_kernel(pmc_maf_t m, const float* __restrict__ in,
                                                              float* __restrict__ out,
                                                              float* __restrict__ ladj_out, int64_t n,
                                                              long long* __restrict__ prof) {
    long long c_setup = 0, c_top = 0, c_burst = 0, c_stage = 0, c_r = 0, c_chain = 0, c_tail = 0;
    const long long c_begin = TICK3();
    long long tk = c_begin;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int q = lane >> 4, p = lane & 15;
    const int64_t row0 = (int64_t)blockIdx.x * 16;
    const int D = m.D, Dp = m.Dp, Hp = m.Hp, T = m.T, nT = m.nT, nXT = m.nXT, nOT = m.nOT;
    float* Y = smem;
    float* X = Y + Dp * 16;
    float* H0 = X + Dp * 16;
    float* H1 = H0 + Hp * 16;
    float* H2 = H1 + Hp * 16;
    float* S = H2 + Hp * 16;                   // staging: [3 layers][16 p][16 rows] then [MAXO][16 p][16 rows]
    float* SO = S + 3 * 256;
    const int* feat_of_rank = m.meta + 8;
    const int* rank_of_feat = m.meta + 8 + T * D;
    const int* quad_meta = m.meta + 8 + 2 * T * D;

    load_rows(Y, in, row0, n, D, Dp, feat_of_rank + (T - 1) * D, lane);
    {   // padding slots of the activations are read by the bursts (times zero weights): zero once
        float4* z4 = reinterpret_cast<float4*>(H0);
        const int n4 = (3 * Hp * 16) >> 2;
        for (int e = lane; e < n4; e += 64) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float ladj = 0.0f;                         // replicated: every lane accumulates its walker's sum
    // lane offsets of the R-layout gathers inside a natural 64-lane fragment record
    const int rl_hidden = (q << 4) + (lane & 3);          // + 4*jt

    for (int t = T - 1; t >= 0; --t) {
        const MafView w = maf_view(m, t);
        {
            float4* z4 = reinterpret_cast<float4*>(X);
            for (int e = lane; e < (Dp * 16) >> 2; e += 64) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();

        Chain<MAXO> s;
        LAP(c_tail)
#pragma unroll
        for (int O = 0; O < MAXO; ++O) {
            const float4 bb = (O < nOT) ? *reinterpret_cast<const float4*>(w.b3 + 16 * O + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            s.oN[O][0] = bb.x; s.oN[O][1] = bb.y; s.oN[O][2] = bb.z; s.oN[O][3] = bb.w;
        }
        // ---- rank 0 reads nothing: bias only
        {
            const float shift = w.b3[0], ls = fast_ls(w.b3[1]);
            const float xv = (Y[lidx(0, p)] - shift) * fast_exp_neg(ls);
            ladj -= ls;
            if (q == 0) X[lidx(0, p)] = xv;
        }
        WAVE_LDS_FENCE();

        // prefetch registers of the next tile's bursts
        float4 pf0[PX3], pf1[PK3], pf2[PK3], pb0, pb1, pb2;
#define PREFETCH3(TT)                                                                                       \
        {                                                                                                   \
            const int TT_ = (TT);                                                                           \
            const float4* f0_ = w.f0 + ((size_t)TT_ * nXT) * 64 + lane;                                     \
            _Pragma("unroll") for (int i_ = 0; i_ < PX3; ++i_) if (i_ < nXT) pf0[i_] = f0_[i_ * 64];        \
            const float4* f1_ = w.f1 + ((size_t)TT_ * nT) * 64 + lane;                                      \
            const float4* f2_ = w.f2 + ((size_t)TT_ * nT) * 64 + lane;                                      \
            _Pragma("unroll") for (int i_ = 0; i_ < PK3; ++i_) if (i_ < TT_) { pf1[i_] = f1_[i_ * 64]; pf2[i_] = f2_[i_ * 64]; } \
            pb0 = *reinterpret_cast<const float4*>(w.b0 + 16 * TT_ + 4 * q);                                \
            pb1 = *reinterpret_cast<const float4*>(w.b1 + 16 * TT_ + 4 * q);                                \
            pb2 = *reinterpret_cast<const float4*>(w.b2 + 16 * TT_ + 4 * q);                                \
        }
        PREFETCH3(0);
        NextFrags<MAXO> nx;
        LOAD_CHAIN_FRAGS(nx, 0);
        LAP(c_setup)

        for (int Tt = 0; Tt < nT; ++Tt) {
            const int4 dg = nx.dg;
            if (dg.x >= D && dg.y >= D && dg.z >= D && dg.w >= D) break;       // padding tiles
            const int pat = 1 | ((dg.y != dg.x) << 1) | ((dg.z != dg.y) << 2) | ((dg.w != dg.z) << 3);

            // ---- this tile's chain fragments were fetched while the previous tile ran
            s.g[0] = nx.g[0]; s.g[1] = nx.g[1]; s.g[2] = nx.g[2]; s.g[3] = nx.g[3];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) { s.wd1[jt] = nx.wd1[jt]; s.wd2[jt] = nx.wd2[jt]; }
            s.wo[0] = nx.wo[0]; s.wo[1] = nx.wo[1];
#pragma unroll
            for (int O = 0; O < MAXO; ++O) s.f3n[O] = nx.f3n[O];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int jt = i + 1; jt < 4; ++jt) s.w0r[i][jt] = nx.w0r[i][jt];
#pragma unroll
            for (int i = 0; i < 4; ++i) s.yv[i] = Y[lidx(s.g[i] < D ? s.g[i] : 0, p)];
            LAP(c_top)
            // ---- natural-layout bursts against everything that is already final
            f32x4 a0, a1, a2;
            a0[0] = pb0.x; a0[1] = pb0.y; a0[2] = pb0.z; a0[3] = pb0.w;
            a1[0] = pb1.x; a1[1] = pb1.y; a1[2] = pb1.z; a1[3] = pb1.w;
            a2[0] = pb2.x; a2[1] = pb2.y; a2[2] = pb2.z; a2[3] = pb2.w;
#pragma unroll
            for (int i = 0; i < PX3; ++i) {
                if (i < nXT) {
                    const float4 b = *reinterpret_cast<const float4*>(X + (i << 8) + (lane << 2));
                    a0 = MFMA(pf0[i].x, b.x, a0); a0 = MFMA(pf0[i].y, b.y, a0);
                    a0 = MFMA(pf0[i].z, b.z, a0); a0 = MFMA(pf0[i].w, b.w, a0);
                }
            }
            for (int Xt = PX3; Xt < nXT; ++Xt) a0 = tile_mac(a0, w.f0 + (size_t)Tt * nXT * 64, X, Xt, lane);
#pragma unroll
            for (int i = 0; i < PK3; ++i) {
                if (i < Tt) {
                    const float4 b1 = *reinterpret_cast<const float4*>(H0 + (i << 8) + (lane << 2));
                    const float4 b2 = *reinterpret_cast<const float4*>(H1 + (i << 8) + (lane << 2));
                    a1 = MFMA(pf1[i].x, b1.x, a1); a2 = MFMA(pf2[i].x, b2.x, a2);
                    a1 = MFMA(pf1[i].y, b1.y, a1); a2 = MFMA(pf2[i].y, b2.y, a2);
                    a1 = MFMA(pf1[i].z, b1.z, a1); a2 = MFMA(pf2[i].z, b2.z, a2);
                    a1 = MFMA(pf1[i].w, b1.w, a1); a2 = MFMA(pf2[i].w, b2.w, a2);
                }
            }
            for (int K = PK3; K < Tt; ++K) {
                a1 = tile_mac(a1, w.f1 + (size_t)Tt * nT * 64, H0, K, lane);
                a2 = tile_mac(a2, w.f2 + (size_t)Tt * nT * 64, H1, K, lane);
            }
            if (PROF) asm volatile("s_nop 0" :: "v"(a0[0]), "v"(a1[0]), "v"(a2[0]));
            LAP(c_burst)
            // ---- stage natural -> R layout ([p][row] so that a quad is one float4)
            {
                float* sp = S + (p << 4) + (q << 2);
                *reinterpret_cast<float4*>(sp) = make_float4(a0[0], a0[1], a0[2], a0[3]);
                *reinterpret_cast<float4*>(sp + 256) = make_float4(a1[0], a1[1], a1[2], a1[3]);
                *reinterpret_cast<float4*>(sp + 512) = make_float4(a2[0], a2[1], a2[2], a2[3]);
                float* so = SO + (p << 4) + (q << 2);
#pragma unroll
                for (int O = 0; O < MAXO; ++O)
                    *reinterpret_cast<float4*>(so + O * 256) = make_float4(s.oN[O][0], s.oN[O][1], s.oN[O][2], s.oN[O][3]);
            }
            if (Tt + 1 < nT) { PREFETCH3(Tt + 1); LOAD_CHAIN_FRAGS(nx, Tt + 1); }   // overlap the chain below
            WAVE_LDS_FENCE();

            LAP(c_stage)
            // ---- staged pre-activations -> R layout
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                const float4 v0 = *reinterpret_cast<const float4*>(S + (p << 4) + (jt << 2));
                const float4 v1 = *reinterpret_cast<const float4*>(S + 256 + (p << 4) + (jt << 2));
                const float4 v2 = *reinterpret_cast<const float4*>(S + 512 + (p << 4) + (jt << 2));
                s.a0[jt][0] = v0.x; s.a0[jt][1] = v0.y; s.a0[jt][2] = v0.z; s.a0[jt][3] = v0.w;
                s.a1[jt][0] = v1.x; s.a1[jt][1] = v1.y; s.a1[jt][2] = v1.z; s.a1[jt][3] = v1.w;
                s.a2[jt][0] = v2.x; s.a2[jt][1] = v2.y; s.a2[jt][2] = v2.z; s.a2[jt][3] = v2.w;
            }
            s.outR[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            s.outR[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gg = s.g[i] < D ? s.g[i] : 0;
                // partial (shift, raw) of rank gg from all previous tiles: rows 2*(gg&7), +1 of output tile gg>>3
                s.po[i] = *reinterpret_cast<const float2*>(SO + (gg >> 3) * 256 + (p << 4) + 2 * (gg & 7));
            }

            if (PROF) asm volatile("s_nop 0" :: "v"(s.a0[0][0]), "v"(s.po[3].x), "v"(s.wd1[0].x), "v"(s.wo[0].x), "v"(s.f3n[0].x));
            LAP(c_r)
            switch (pat) {
#define CASE(P) case P: chain_group<P, 0, MAXO>(s, H0, H1, H2, X, Y, Tt, D, nOT, q, p, ladj); break;
                CASE(1) CASE(3) CASE(5) CASE(7) CASE(9) CASE(11) CASE(13) CASE(15)
#undef CASE
            }
            WAVE_LDS_FENCE();
            LAP(c_chain)
        }
#undef PREFETCH3

        __syncthreads();
        const bool last = (t == 0);
        rerank_or_store(X, Y, out, row0, n, D, Dp, feat_of_rank + t * D,
                        last ? nullptr : rank_of_feat + (t - 1) * D, lane);
        __syncthreads();
    }
    if (ladj_out && lane < 16 && row0 + p < n) ladj_out[row0 + p] = ladj;
    if (PROF && lane == 0) {
        LAP(c_tail)
        long long* P = prof + (size_t)blockIdx.x * 8;
        P[0] = TICK3() - c_begin; P[1] = c_setup; P[2] = c_top; P[3] = c_burst; P[4] = c_stage; P[5] = c_r;
        P[6] = c_chain; P[7] = c_tail;
    }
}

extern "C" int pmc_debug_inverse3_profile(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n,
                                          long long* prof, void* stream) {
    const size_t lds = (size_t)(2 * m->Dp * 16 + 3 * m->Hp * 16 + 3 * 256 + 4 * 256) * sizeof(float);
    hipLaunchKernelGGL((maf_inverse_tri3_kernel<4, true>), dim3((unsigned)((n + 15) / 16)), dim3(64), lds,
                       (hipStream_t)stream, *m, z, x, ladj, n, prof);
    return pmc_check_launch("maf_inverse_tri3_kernel<prof>");
}

int pmc_launch_inverse_tri3(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n, hipStream_t stream) {
    if (m->nOT > 8) return -1;                                     // caller falls back to the tri2 sweep
    const int maxo = m->nOT <= 4 ? 4 : 8;
    const size_t lds = (size_t)(2 * m->Dp * 16 + 3 * m->Hp * 16 + 3 * 256 + maxo * 256) * sizeof(float);
    if (lds > 160 * 1024) return -1;
#define LAUNCH(MO)                                                                                               \
    {                                                                                                            \
        if (lds > 48 * 1024) {                                                                                   \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(maf_inverse_tri3_kernel<MO, false>),        \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
            if (e != hipSuccess) return pmc_fail_hip(e, "hipFuncSetAttribute(maf_inverse_tri3_kernel)");         \
        }                                                                                                        \
        hipLaunchKernelGGL((maf_inverse_tri3_kernel<MO, false>), dim3((unsigned)((n + 15) / 16)), dim3(64), lds,   \
                           stream, *m, z, x, ladj, n, (long long*)nullptr);                                      \
    }
    if (maxo == 4) LAUNCH(4) else LAUNCH(8)
#undef LAUNCH
    return pmc_check_launch("maf_inverse_tri3_kernel");
}

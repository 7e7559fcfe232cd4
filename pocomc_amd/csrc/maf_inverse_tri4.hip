// Triangular-sweep MAF inverse, register-resident chain (maf_chain.h) + buffer-addressed weights.
//
// PMC on maf_inverse_tri3_kernel (profiles/r01_c_rocprof_summary.txt): 15.7 k VALU instructions against
// 2061 MFMAs per wave and the MFMA pipe busy only 26 % of the wave's lifetime -- most VALU work was
// 64-bit address arithmetic for ~50 weight loads per tile and the copies of a double-buffered
// fragment set.  This version
//   * reads every weight through ONE bounds-checked buffer resource per transform
//     (raw_buffer_load_b128: wave-constant VGPR offset + SGPR byte offset + immediate), so a load
//     costs one SALU add instead of 5-6 VALU instructions and needs no per-array base pointers;
//   * fetches the chain's own fragments at the top of the tile, where the natural-layout bursts
//     (tens of MFMAs) hide their latency -- no second register set, no copies;
//   * keeps the prefetch of the NEXT tile's burst fragments issued right after the bursts.

#include <stdlib.h>
#include "maf_chain_rot.h"

#define DG_WORDS(m) ((m)->nT * 4 + 8)      // LDS words of the tiles' degree table (4 per tile, padded)
// + the two-wave sweep's permutation / rank-0 tables and its second x array (with alignment slack)
#define TRI5_TT_WORDS(m) (((m)->nT + 2) * 16)  // the two-wave sweep's per-tile table (16 words per hidden tile, two rows of "no groups" behind)
#define TRI5_YT_WORDS(m) ((m)->T * (((m)->nT + 2) * 4 + 1))   // per transform: the y offsets of every tile's groups, of rank 0
#define TRI5_TABLE_WORDS(m) (((TRI5_TT_WORDS(m) + (m)->Dp + 2 * (m)->T + 3) & ~3) + (m)->Dp * 16 + ((TRI5_YT_WORDS(m) + 3) & ~3))
#define TRI5_LDS_FLOATS(m, maxo) (2 * (m)->Dp * 16 + 2 * (m)->Hp * 16 + 2 * 256 + 2 * (3 + (maxo)) * 256 + TRI5_TABLE_WORDS(m))
#include "propose_body.h"

#define PX4 2
#define PK4 8


// Straight-line burst of hidden tile TT against tiles 0..TT-1 (TT is a compile-time constant):
// every LDS read is issued up front and two independent accumulators per layer keep the MFMAs
// back to back (a dependent v_mfma_f32_16x16x4_f32 waits 40 cycles, an independent one issues after 32).
// (burst_tile_open leaves the two partial sums apart: the two-wave kernel adds its last K tile before folding them,
// so that both kernels round alike)
template <int TT>
__device__ __forceinline__ void burst_tile_open(f32x4& a1, f32x4& a2, f32x4& c1, f32x4& c2, const float4 (&pf1)[PK4],
                                                const float4 (&pf2)[PK4], const float* H0, const float* H1, int lane) {
    if constexpr (TT > 0) {
        float4 b1[TT], b2[TT];
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            b1[i] = *reinterpret_cast<const float4*>(H0 + (i << 8) + (lane << 2));
            b2[i] = *reinterpret_cast<const float4*>(H1 + (i << 8) + (lane << 2));
        }
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            a1 = MFMA(pf1[i].x, b1[i].x, a1); a2 = MFMA(pf2[i].x, b2[i].x, a2);
            c1 = MFMA(pf1[i].y, b1[i].y, c1); c2 = MFMA(pf2[i].y, b2[i].y, c2);
            a1 = MFMA(pf1[i].z, b1[i].z, a1); a2 = MFMA(pf2[i].z, b2[i].z, a2);
            c1 = MFMA(pf1[i].w, b1[i].w, c1); c2 = MFMA(pf2[i].w, b2[i].w, c2);
        }
    }
}

template <int TT>
__device__ __forceinline__ void burst_tile(f32x4& a1, f32x4& a2, const float4 (&pf1)[PK4], const float4 (&pf2)[PK4],
                                           const float* H0, const float* H1, int lane) {
    if constexpr (TT > 0) {
        float4 b1[TT], b2[TT];
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            b1[i] = *reinterpret_cast<const float4*>(H0 + (i << 8) + (lane << 2));
            b2[i] = *reinterpret_cast<const float4*>(H1 + (i << 8) + (lane << 2));
        }
        f32x4 c1 = {0.f, 0.f, 0.f, 0.f}, c2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            a1 = MFMA(pf1[i].x, b1[i].x, a1); a2 = MFMA(pf2[i].x, b2[i].x, a2);
            c1 = MFMA(pf1[i].y, b1[i].y, c1); c2 = MFMA(pf2[i].y, b2[i].y, c2);
            a1 = MFMA(pf1[i].z, b1[i].z, a1); a2 = MFMA(pf2[i].z, b2[i].z, a2);
            c1 = MFMA(pf1[i].w, b1[i].w, c1); c2 = MFMA(pf2[i].w, b2[i].w, c2);
        }
        for (int r = 0; r < 4; ++r) { a1[r] += c1[r]; a2[r] += c2[r]; }
    }
}

// the two-wave sweep's form: one accumulator per layer, the two layers alternating (a dependent MFMA is two issue slots
// behind its predecessor)
template <int TT>
__device__ __forceinline__ void burst_tile_chain(f32x4& a1, f32x4& a2, const float4 (&pf1)[PK4], const float4 (&pf2)[PK4],
                                                 const float* H0, const float* H1, int lane) {
    if constexpr (TT > 0) {
        float4 b1[TT], b2[TT];
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            b1[i] = *reinterpret_cast<const float4*>(H0 + (i << 8) + (lane << 2));
            b2[i] = *reinterpret_cast<const float4*>(H1 + (i << 8) + (lane << 2));
        }
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            a1 = MFMA(pf1[i].x, b1[i].x, a1); a2 = MFMA(pf2[i].x, b2[i].x, a2);
            a1 = MFMA(pf1[i].y, b1[i].y, a1); a2 = MFMA(pf2[i].y, b2[i].y, a2);
            a1 = MFMA(pf1[i].z, b1[i].z, a1); a2 = MFMA(pf2[i].z, b2[i].z, a2);
            a1 = MFMA(pf1[i].w, b1[i].w, a1); a2 = MFMA(pf2[i].w, b2[i].w, a2);
        }
    }
}

template <int TT>
__device__ __forceinline__ void prefetch_tile(float4 (&pf1)[PK4], float4 (&pf2)[PK4], __amdgpu_buffer_rsrc_t rs,
                                              int vo_lane, int so1, int so2) {
#pragma unroll
    for (int i = 0; i < TT; ++i) {
        pf1[i] = bload4(rs, vo_lane, so1 + i * 1024);
        pf2[i] = bload4(rs, vo_lane, so2 + i * 1024);
    }
}

// ABL: timing-only ablations (see maf_chain_rot.h); 4 = no bursts, 8 = no chain, 16 = no next-tile prefetch,
// 32 = no tile-top fragment loads.  FM: 0 = plain inverse of `in`; 4 / 8 / 16 = fused proposal with D <= 4 FM.
template <int MAXO, int ABL, int FM = 0>
__global__ __launch_bounds__(64) void maf_inverse_tri4_kernel(pmc_maf_t m, const float* __restrict__ in,
                                                              float* __restrict__ out,
                                                              float* __restrict__ ladj_out, int64_t n,
                                                              ProposeArgs pa) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int q = lane >> 4, p = lane & 15;
    const int64_t row0 = (int64_t)blockIdx.x * 16;
    const int D = m.D, Dp = m.Dp, Hp = m.Hp, T = m.T, nT = m.nT, nXT = m.nXT, nOT = m.nOT;
    float* Y = smem;
    float* X = Y + Dp * 16;
    float* H0 = X + Dp * 16;
    float* H1 = H0 + Hp * 16;
    float* S = H1 + Hp * 16;                   // staging: [3 layers][16 p][16 rows] then [MAXO][16 p][16 rows]
    float* SO = S + 3 * 256;
    const int* feat_of_rank = m.meta + 8;
    const int* rank_of_feat = m.meta + 8 + T * D;
    const int* quad_meta = m.meta + 8 + 2 * T * D;

    // byte offsets of the packed arrays inside one transform's block (maf_spec.py: pk_offsets)
    const int oF0 = 0;
    const int oF1 = oF0 + nT * nXT * 1024;
    const int oF2 = oF1 + nT * nT * 1024;
    const int oF3 = oF2 + nT * nT * 1024;
    const int oW0 = oF3 + nOT * nT * 1024;
    const int oB0 = oW0 + Dp * Hp * 4;
    const int oB1 = oB0 + Hp * 4;
    const int oB2 = oB1 + Hp * 4;
    const int oB3 = oB2 + Hp * 4;
    const int blk_bytes = (int)(m.pk_per_transform * 4);

    // wave-constant lane offsets (bytes)
    const int vo_lane = lane << 4;                              // natural fragment record: lane * 16 B
    const int vo_T = chain_vo_T(lane);                          // transposed gather inside a record (maf_chain_rot.h)
    const int vo_q = q << 4;                                    // 4 consecutive floats of quad q

    if constexpr (FM > 0) {
        for (int e = lane; e < (Dp - D) * 16; e += 64) Y[lidx(D + (e >> 4), e & 15)] = 0.0f;
        const double sg = pa.adapt ? pa.adapt[0] : pa.sigma, ca = pa.adapt ? pa.adapt[1] : pa.cn_a;
        propose_body<FM>(pa.kind, pa.cur32, nullptr, pa.adapt ? pa.adapt + 2 : pa.mu, pa.inv_cov, pa.chol, pa.nu, sg, ca,
                         pa.rng, pa.prop64, nullptr, pa.quad, pa.quad_prop, n, D, Y, rank_of_feat + (T - 1) * D);
    } else {
        load_rows(Y, in, row0, n, D, Dp, feat_of_rank + (T - 1) * D, lane);
    }
    {   // padding slots of the activations are read by the bursts (times zero weights): zero once
        float4* z4 = reinterpret_cast<float4*>(H0);
        const int n4 = (2 * Hp * 16) >> 2;
        for (int e = lane; e < n4; e += 64) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // the tiles' degree words in LDS: read from global memory in the middle of a tile they cost the chain a whole L2
    // round trip (the wait also covers the fragment prefetches issued just before)
    int* DGT = reinterpret_cast<int*>(SO + MAXO * 256);
    for (int e = lane; e < nT * 4; e += 64) DGT[e] = quad_meta[e];
    __syncthreads();
    float ladj = 0.0f;

    for (int t = T - 1; t >= 0; --t) {
        const float* blk = m.packed + (size_t)t * m.pk_per_transform;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)blk, 0, blk_bytes, 0x00020000);
        {
            float4* z4 = reinterpret_cast<float4*>(X);
            for (int e = lane; e < (Dp * 16) >> 2; e += 64) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();

        ChainRot<MAXO> s;
        ChainFrags<MAXO> f;
#pragma unroll
        for (int O = 0; O < MAXO; ++O) {
            const float4 bb = (O < nOT) ? bload4(rs, vo_q, oB3 + 64 * O) : make_float4(0.f, 0.f, 0.f, 0.f);
            s.oN[O][0] = bb.x; s.oN[O][1] = bb.y; s.oN[O][2] = bb.z; s.oN[O][3] = bb.w;
        }
        // ---- rank 0 reads nothing: bias only
        {
            const float* b3 = blk + (oB3 >> 2);
            const float shift = b3[0], ls = fast_ls(b3[1]);
            const float xv = (Y[lidx(0, p)] - shift) * fast_exp_neg(ls);
            ladj -= ls;
            if (q == 0) X[lidx(0, p)] = xv;
        }
        WAVE_LDS_FENCE();

        // burst fragments of the NEXT tile (filled while the current tile's chain runs)
        float4 pf0[PX4], pf1[PK4], pf2[PK4], pb0, pb1, pb2;
#define PREFETCH4(TT)                                                                                          \
        if (!(ABL & 16)) {                                                                                                      \
            const int TT_ = (TT);                                                                              \
            _Pragma("unroll") for (int i_ = 0; i_ < PX4; ++i_)                                                 \
                if (i_ < nXT) pf0[i_] = bload4(rs, vo_lane, oF0 + (TT_ * nXT + i_) * 1024);                     \
            {                                                                                                  \
                const int so1_ = oF1 + TT_ * nT * 1024, so2_ = oF2 + TT_ * nT * 1024;                          \
                switch (TT_ < PK4 ? TT_ : PK4) {                                                               \
                    case 1: prefetch_tile<1>(pf1, pf2, rs, vo_lane, so1_, so2_); break;                        \
                    case 2: prefetch_tile<2>(pf1, pf2, rs, vo_lane, so1_, so2_); break;                        \
                    case 3: prefetch_tile<3>(pf1, pf2, rs, vo_lane, so1_, so2_); break;                        \
                    case 4: prefetch_tile<4>(pf1, pf2, rs, vo_lane, so1_, so2_); break;                        \
                    case 5: prefetch_tile<5>(pf1, pf2, rs, vo_lane, so1_, so2_); break;                        \
                    case 6: prefetch_tile<6>(pf1, pf2, rs, vo_lane, so1_, so2_); break;                        \
                    case 7: prefetch_tile<7>(pf1, pf2, rs, vo_lane, so1_, so2_); break;                        \
                    case 8: prefetch_tile<8>(pf1, pf2, rs, vo_lane, so1_, so2_); break;                        \
                    default: break;                                                                            \
                }                                                                                              \
            }                                                                                                  \
            pb0 = bload4(rs, vo_q, oB0 + 64 * TT_);                                                            \
            pb1 = bload4(rs, vo_q, oB1 + 64 * TT_);                                                            \
            pb2 = bload4(rs, vo_q, oB2 + 64 * TT_);                                                            \
        }
        PREFETCH4(0);
        int4 dg_next = *reinterpret_cast<const int4*>(DGT);

        for (int Tt = 0; Tt < nT; ++Tt) {
            int4 dg = dg_next;
            dg.x &= 0xffff; dg.y &= 0xffff; dg.z &= 0xffff; dg.w &= 0xffff;
            if (dg.x >= D && dg.y >= D && dg.z >= D && dg.w >= D) break;       // padding tiles
            const int pat = 1 | ((dg.y != dg.x) << 1) | ((dg.z != dg.y) << 2) | ((dg.w != dg.z) << 3);

            // ---- the chain's own fragments: issued first, they land while the bursts run
            {
                const bool ny = dg.y != dg.x, nz = dg.z != dg.y, nw = dg.w != dg.z;
                f.g[0] = dg.x;
                f.g[1] = ny ? dg.y : (nz ? dg.z : (nw ? dg.w : D));
                f.g[2] = ny ? (nz ? dg.z : (nw ? dg.w : D)) : ((nz && nw) ? dg.w : D);
                f.g[3] = (ny && nz && nw) ? dg.w : D;
            }
            const int soD1 = oF1 + (Tt * nT + Tt) * 1024, soD2 = oF2 + (Tt * nT + Tt) * 1024;
            if (!(ABL & 32)) {
            f.wt1 = bload4(rs, vo_T, soD1);
            f.wt2 = bload4(rs, vo_T, soD2);
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                // rows (lane&3): 0,1 -> (shift, raw) of group 2*sl; 2,3 -> group 2*sl+1
                const int g_even = f.g[2 * sl], g_odd = f.g[2 * sl + 1];
                const int gsel = (lane & 2) ? g_odd : g_even;
                const bool ok = gsel < D;
                const int gg = ok ? gsel : 0;
                const int vo = ((((gg >> 3) * nT) << 6) + (q << 4) + 2 * (gg & 7) + (lane & 1)) << 4;
                const float4 v = bload4(rs, vo, oF3 + Tt * 1024);
                f.wo[sl] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int O = 0; O < MAXO; ++O)
                f.f3n[O] = (O < nOT) ? bload4(rs, vo_lane, oF3 + (O * nT + Tt) * 1024) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int gg = f.g[i] < D ? f.g[i] : 0;
                float w[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int jt = i + 1; jt < 4; ++jt)
                    w[jt] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        rs, q << 2, oW0 + (gg * Hp + 16 * Tt + 4 * jt) * 4, 0));
                f.w0o[i] = make_float4(w[0], w[1], w[2], w[3]);
            }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) s.yv[i] = Y[lidx(f.g[i] < D ? f.g[i] : 0, p)];

            // ---- natural-layout bursts against everything that is already final
            f32x4 a0, a1, a2;
            a0[0] = pb0.x; a0[1] = pb0.y; a0[2] = pb0.z; a0[3] = pb0.w;
            a1[0] = pb1.x; a1[1] = pb1.y; a1[2] = pb1.z; a1[3] = pb1.w;
            a2[0] = pb2.x; a2[1] = pb2.y; a2[2] = pb2.z; a2[3] = pb2.w;
#pragma unroll
            for (int i = 0; i < PX4; ++i) {
                if (i < nXT) {
                    const float4 b = *reinterpret_cast<const float4*>(X + (i << 8) + (lane << 2));
                    a0 = MFMA(pf0[i].x, b.x, a0); a0 = MFMA(pf0[i].y, b.y, a0);
                    a0 = MFMA(pf0[i].z, b.z, a0); a0 = MFMA(pf0[i].w, b.w, a0);
                }
            }
            for (int Xt = PX4; Xt < nXT; ++Xt) {
                const float4 a = bload4(rs, vo_lane, oF0 + (Tt * nXT + Xt) * 1024);
                const float4 b = *reinterpret_cast<const float4*>(X + (Xt << 8) + (lane << 2));
                a0 = MFMA(a.x, b.x, a0); a0 = MFMA(a.y, b.y, a0); a0 = MFMA(a.z, b.z, a0); a0 = MFMA(a.w, b.w, a0);
            }
            if (!(ABL & 4))
            switch (Tt < PK4 ? Tt : PK4) {
                case 1: burst_tile<1>(a1, a2, pf1, pf2, H0, H1, lane); break;
                case 2: burst_tile<2>(a1, a2, pf1, pf2, H0, H1, lane); break;
                case 3: burst_tile<3>(a1, a2, pf1, pf2, H0, H1, lane); break;
                case 4: burst_tile<4>(a1, a2, pf1, pf2, H0, H1, lane); break;
                case 5: burst_tile<5>(a1, a2, pf1, pf2, H0, H1, lane); break;
                case 6: burst_tile<6>(a1, a2, pf1, pf2, H0, H1, lane); break;
                case 7: burst_tile<7>(a1, a2, pf1, pf2, H0, H1, lane); break;
                case 8: burst_tile<8>(a1, a2, pf1, pf2, H0, H1, lane); break;
                default: break;
            }
            if (Tt > PK4) {                                 // flows wider than PK4 tiles: four K tiles' fragments in flight
                float4 w1r[4], w2r[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int Kl = min(PK4 + j, nT - 1);
                    w1r[j] = bload4(rs, vo_lane, oF1 + (Tt * nT + Kl) * 1024);
                    w2r[j] = bload4(rs, vo_lane, oF2 + (Tt * nT + Kl) * 1024);
                }
                for (int K0 = PK4; K0 < Tt; K0 += 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int K = K0 + j;
                        const float4 w1 = w1r[j], w2 = w2r[j];
                        const int Kn = min(K + 4, nT - 1);
                        w1r[j] = bload4(rs, vo_lane, oF1 + (Tt * nT + Kn) * 1024);
                        w2r[j] = bload4(rs, vo_lane, oF2 + (Tt * nT + Kn) * 1024);
                        if (K < Tt) {
                            const float4 b1 = *reinterpret_cast<const float4*>(H0 + (K << 8) + (lane << 2));
                            const float4 b2 = *reinterpret_cast<const float4*>(H1 + (K << 8) + (lane << 2));
                            a1 = MFMA(w1.x, b1.x, a1); a2 = MFMA(w2.x, b2.x, a2);
                            a1 = MFMA(w1.y, b1.y, a1); a2 = MFMA(w2.y, b2.y, a2);
                            a1 = MFMA(w1.z, b1.z, a1); a2 = MFMA(w2.z, b2.z, a2);
                            a1 = MFMA(w1.w, b1.w, a1); a2 = MFMA(w2.w, b2.w, a2);
                        }
                    }
                }
            }
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
            WAVE_LDS_FENCE();
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                // only row q of every quad is ever used by this lane (rotated layout)
                s.a0[jt] = S[(p << 4) + (jt << 2) + q];
                s.p1[jt] = S[256 + (p << 4) + (jt << 2) + q];
                s.p2[jt] = S[512 + (p << 4) + (jt << 2) + q];
            }
            s.acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
            s.acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
            s.outR[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            s.outR[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gg = f.g[i] < D ? f.g[i] : 0;
                s.po[i] = *reinterpret_cast<const float2*>(SO + (gg >> 3) * 256 + (p << 4) + 2 * (gg & 7));
            }

            // first group, then the next tile's burst fragments (loads return in order: issued any earlier
            // they would sit between the chain and its own fragments), then the remaining groups
            chain_tile_begin(s, f, X, S, D, q, p, lane);
            if (!(ABL & 8))
            switch (pat) {
#define CASE(P) case P: chain_group_rot<P, 0, 1, MAXO, ABL>(s, f, H0, H1, X, Tt, D, nOT, q, p, ladj); break;
                CASE(1) CASE(3) CASE(5) CASE(7) CASE(9) CASE(11) CASE(13) CASE(15)
#undef CASE
            }
            if (Tt + 1 < nT) {
                PREFETCH4(Tt + 1);
                dg_next = *reinterpret_cast<const int4*>(DGT + 4 * (Tt + 1));
            }
            if (!(ABL & 8))
            switch (pat) {
#define CASE(P) case P: chain_group_rot<P, 1, 4, MAXO, ABL>(s, f, H0, H1, X, Tt, D, nOT, q, p, ladj); break;
                CASE(3) CASE(5) CASE(7) CASE(9) CASE(11) CASE(13) CASE(15)
#undef CASE
                default: break;
            }
            chain_flush(s, ladj);
            WAVE_LDS_FENCE();
        }
#undef PREFETCH4
        __syncthreads();
        const bool last = (t == 0);
        rerank_or_store(X, Y, out, row0, n, D, Dp, feat_of_rank + t * D,
                        last ? nullptr : rank_of_feat + (t - 1) * D, lane);
        __syncthreads();
    }
    if (ladj_out && lane < 16 && row0 + p < n) ladj_out[row0 + p] = ladj;
    if constexpr (FM > 0) {
        // the scaler (+ prior) on the 16 walkers while they are in LDS (X: by rank of the first transform); the
        // activation tiles are free now and serve as its scratch
        if (pa.epi.on)
            scaler_epilogue(pa.epi, X, rank_of_feat, reinterpret_cast<double*>(H0), row0, n, D, lane, 64,
                            [](int r, int pp) { return lidx(r, pp); });
    }
}

static bool tri5_wanted(const pmc_maf_t* m, int64_t n);
static int launch_tri5(const ProposeArgs* pa, const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n,
                       hipStream_t stream);

// PMC_INVERSE_LANE=1: AUTO takes the lane-per-walker sweep (maf_inverse_tri6.hip) wherever it covers the flow (A/B
// runs).  Default: the register-chain sweeps of this file for flows of < 16 hidden tiles (maf3 @ D = 32: 61-64 us
// against 66-83 us for up to 8192 rows), the lane-per-walker sweep for the wider ones (pmc_tri6_preferred: D = 50 / maf6
// 314 against 650 us, D = 128 / 8 transforms 0.69 ms per round) and for everything with more than 8 output tiles.
static bool lane_sweep_enabled() {
    static const bool on = pmc_env_int("PMC_INVERSE_LANE", 0) != 0;
    return on;
}

int pmc_launch_inverse_tri4(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n, hipStream_t stream,
                            int variant) {
    if (variant < 0 && (lane_sweep_enabled() || pmc_tri6_preferred(m))) {
        const int rc = pmc_launch_tri6(nullptr, m, z, x, ladj, n, stream);
        if (rc >= 0) return rc;
    }
    if (variant == 1) return launch_tri5(nullptr, m, z, x, ladj, n, stream);
    if (variant < 0 && tri5_wanted(m, n)) {
        const int rc = launch_tri5(nullptr, m, z, x, ladj, n, stream);
        if (rc >= 0) return rc;
    }
    if (m->pk_per_transform * 4 > 0x7fffffffLL) return -1;         // 32-bit buffer offsets
    const int maxo = m->nOT <= 4 ? 4 : 8;
    const size_t lds = (size_t)(2 * m->Dp * 16 + 2 * m->Hp * 16 + 3 * 256 + maxo * 256 + DG_WORDS(m)) * sizeof(float);
    if (m->nOT > 8 || lds > 160 * 1024) {
        // wide flows (D > 64): lane-per-walker sweep; -1 if that does not cover the flow either (caller falls back)
        return variant < 0 ? pmc_launch_tri6(nullptr, m, z, x, ladj, n, stream) : -1;
    }
#define LAUNCH(MO)                                                                                               \
    {                                                                                                            \
        if (lds > 48 * 1024) {                                                                                   \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(maf_inverse_tri4_kernel<MO, 0>),        \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
            if (e != hipSuccess) return pmc_fail_hip(e, "hipFuncSetAttribute(maf_inverse_tri4_kernel)");         \
        }                                                                                                        \
        hipLaunchKernelGGL((maf_inverse_tri4_kernel<MO, 0>), dim3((unsigned)((n + 15) / 16)), dim3(64), lds,      \
                           stream, *m, z, x, ladj, n, ProposeArgs{});                                            \
    }
    if (maxo == 4) LAUNCH(4) else LAUNCH(8)
#undef LAUNCH
    return pmc_check_launch("maf_inverse_tri4_kernel");
}

// Proposal (mcmc.py:77-85) + flow inverse (mcmc.py:88) in one launch; -1 when this flow / size is not covered by
// the fused instances (the caller then launches pmc_propose and pmc_maf_inverse).
int pmc_launch_propose_inverse_tri4(int kind, const float* cur32, const double* mu, const double* inv_cov,
                                    const double* chol, double nu, double sigma, double cn_a, const pmc_rng_t* rng,
                                    double* prop64, double* quad, double* quad_prop, const pmc_maf_t* m, float* x,
                                    float* ladj, int64_t n, hipStream_t stream, const double* adapt,
                                    const ScalerEpi* epi, int* epi_done) {
    if (epi_done) *epi_done = 0;
    if (m->n_out == 23) {                                  // spline flows: the two-wave spline sweep has the fused instances
        ProposeArgs pan{kind, cur32, mu, inv_cov, chol, nu, sigma, cn_a, *rng, prop64, quad, quad_prop, adapt};
        return pmc_launch_propose_inverse_nsf2(&pan, epi, epi_done, m, x, ladj, n, stream);
    }
    if (m->n_out != 2 || !m->tri_ok || m->nOT > 8 || m->D > 64) return -1;
    if (m->pk_per_transform * 4 > 0x7fffffffLL) return -1;
    const int maxo = m->nOT <= 4 ? 4 : 8;
    const size_t lds = (size_t)(2 * m->Dp * 16 + 2 * m->Hp * 16 + 3 * 256 + maxo * 256 + DG_WORDS(m)) * sizeof(float);
    if (lds > 160 * 1024) return -1;
    ProposeArgs pa{kind, cur32, mu, inv_cov, chol, nu, sigma, cn_a, *rng, prop64, quad, quad_prop, adapt};
    // the scaler as the sweep's epilogue: its scratch aliases the two activation arrays of the walker set
    const bool epi_ok = epi && epi_done && epi->s.D == m->D && !lane_sweep_enabled() &&
                        scaler_epilogue_lds_bytes(m->D) <= (size_t)2 * m->Hp * 16 * sizeof(float);
    if (epi_ok) { pa.epi = *epi; pa.epi.on = 1; *epi_done = 1; }
    if (lane_sweep_enabled()) {
        const int rc = pmc_launch_tri6(&pa, m, nullptr, x, ladj, n, stream);
        if (rc >= 0) return rc;
    }
    if (tri5_wanted(m, n)) {
        const int rc = launch_tri5(&pa, m, nullptr, x, ladj, n, stream);
        if (rc >= 0) return rc;
    }
#define LAUNCHF(MO, FMV)                                                                                          \
    {                                                                                                             \
        if (lds > 48 * 1024) {                                                                                    \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(maf_inverse_tri4_kernel<MO, 0, FMV>),    \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
            if (e != hipSuccess) return pmc_fail_hip(e, "hipFuncSetAttribute(maf_inverse_tri4_kernel)");          \
        }                                                                                                         \
        hipLaunchKernelGGL((maf_inverse_tri4_kernel<MO, 0, FMV>), dim3((unsigned)((n + 15) / 16)), dim3(64), lds,  \
                           stream, *m, (const float*)nullptr, x, ladj, n, pa);                                    \
    }
    if (m->D <= 16) { if (maxo == 4) LAUNCHF(4, 4) else LAUNCHF(8, 4) }
    else if (m->D <= 32) { if (maxo == 4) LAUNCHF(4, 8) else LAUNCHF(8, 8) }
    else { if (maxo == 4) LAUNCHF(4, 16) else LAUNCHF(8, 16) }
#undef LAUNCHF
    return pmc_check_launch("maf_inverse_tri4_kernel<fused proposal>");
}

// ============================================================================================================
// tri5: the same sweep as TWO wavefronts per 16 walkers -- a CHAIN wave and a BURST wave -- right-looking (round 3).
//
// What bounds the sweep is the latency of one wavefront's in-order instruction stream (maf_chain_rot.h), so the work is
// cut by WHEN its inputs exist, not by flops:
//   * everything a hidden tile Tt needs from tiles <= Tt-2 (and from the ranks they produced) is a dense left-looking
//     product that can be formed a whole tile time ahead: the BURST wave's -- layer 0 against x (the fragment image f0c:
//     W0 with the columns of the ranks of tiles >= Tt-1 zeroed), layers 1 / 2 against h0 / h1, the output rows against h2
//     (once per tile, right-looking, all output tiles that are not identically zero) -- into transposed accumulators
//     (lane (q, p) holds row q of the tile's four quads: exactly what the chain's lane (q, p) adds) and from there into
//     one of two staging buffers in LDS;
//   * what tile Tt needs from tile Tt-1 exists only when that tile's groups have run, i.e. on the CHAIN wave: every group
//     adds, next to its own tile's blocks, its share of the NEXT tile's pre-activations (ChainRot::accN1 / accN2 / outN /
//     a0N: one MFMA per layer, two for the output rows, four FMAs) -- so that at a tile boundary the chain only adds
//     two register sets (staged partial + own share) and goes on; no wave ever waits for work that could not have
//     been started earlier.  The first version of this kernel (round 2) had the burst wave add tile Tt-1's blocks
//     between two barriers while the chain waited, and the chain form the layer-0 product itself: 5.3 k cycles per
//     tile against 2.3 k for the chain's own stream (scripts/micro/chain_tile.hip).
// One LDS-only barrier per tile: E(Tt) = "tile Tt is final, the staging of tile Tt+1 is complete".
// The chain's fragments of tile Tt+1 (16 loads) are requested in the shadows of tile Tt's hops into the other of two
// register set (copied over at the tile boundary), across transform boundaries too.
// Between transforms nothing is re-ranked: transform t reads its y through a per-transform offset table from the x array
// of transform t + 1 (the two x arrays alternate), and the first tile of a transform -- biases only -- is staged while the
// chain still runs the last tile of the transform before ("Transform boundaries").
// Residency: a workgroup lives on one CU (4 SIMDs, one such wave each): 512 walker sets at a time; the launcher
// splits larger calls into rounds of this kernel (two rounds still beat the lone wave's one: DESIGN.md section 4).
// ============================================================================================================
#define TRI5_NC 1                  // chain waves (16-walker sets) per workgroup
#ifndef TRI5_ABL
#define TRI5_ABL 0                 // timing experiments only (scripts/abl_tri5.sh): results are wrong when != 0
#endif
#if TRI5_ABL != 0
extern "C" int pmc_ablation_tri5(void) { return TRI5_ABL; }      // (see pmc_ablation_tri6)
#endif
#define PXB 4                      // x tiles of the layer-0 product held in registers (D <= 64)
#define TRI5_STAGE_FLOATS(MO) ((3 + (MO)) * 256)                 // one staging buffer: S0 | S1 | S2 (transposed, [lane][4]) | SO[MO] (natural)
#define TRI5_SET_FLOATS(Dp, Hp, MO) (2 * (Dp) * 16 + 2 * (Hp) * 16 + 2 * 256 + 2 * TRI5_STAGE_FLOATS(MO))
#define OOB_VOFF 0x40000000        // a lane offset beyond every image: the bounds-checked load returns zeros

template <int MAXO, int FM>
__global__ __launch_bounds__(64 * (TRI5_NC + 1)) void maf_inverse_tri5_kernel(pmc_maf_t m, const float* __restrict__ in,
                                                                              float* __restrict__ out,
                                                                              float* __restrict__ ladj_out, int64_t n,
                                                                              ProposeArgs pa) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (FM > 0) {
        if (pa.epi.on && pa.epi.stamps && threadIdx.x == 0) {          // (measurement only: scaler_body.h, ScalerEpi::stamps)
            pa.epi.stamps[blockIdx.x * 8 + 6] = wall_clock64();
            pa.epi.stamps[blockIdx.x * 8 + 7] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                                ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        }
    }
    const int q = lane >> 4, p = lane & 15;
    const int D = m.D, Dp = m.Dp, Hp = m.Hp, T = m.T, nT = m.nT, nXT = m.nXT, nOT = m.nOT;
    const int nTl = __builtin_amdgcn_readfirstlane(m.meta[7]);                                                   // live hidden tiles (the padding tiles trail)
    const int64_t set = (int64_t)blockIdx.x;
    const int64_t row0 = set * 16;
    float* Y = smem;
    float* XA = Y + Dp * 16;
    float* H0 = XA + Dp * 16;
    float* H1 = H0 + Hp * 16;
    float* H2 = H1 + Hp * 16;                  // h2 of the last two tiles [tile parity][256]: the burst wave's output updates read it
    float* STG = H2 + 2 * 256;                 // two staging buffers (tile parity)
    const int* feat_of_rank = m.meta + 8;
    const int* rank_of_feat = m.meta + 8 + T * D;
    const int* quad_meta = m.meta + 8 + 2 * T * D;

    // byte offsets of the packed arrays inside one transform's block (maf_spec.py: pk_offsets)
    const int oF1 = nT * nXT * 1024;
    const int oF2 = oF1 + nT * nT * 1024;
    const int oF3 = oF2 + nT * nT * 1024;
    const int oW0 = oF3 + nOT * nT * 1024;
    const int oB0 = oW0 + Dp * Hp * 4;
    const int oB3 = oB0 + 3 * Hp * 4;
    const int oCW0 = oB3 + nOT * 64 + 2 * nT * 1024;
    const int oF0C = oCW0 + nT * 1024 + nT * 512;
    const int oB0T = oF0C + nT * nXT * 1024;
    const int oB1T = oB0T + Hp * 4;
    const int oB2T = oB1T + Hp * 4;
    const int blk_bytes = (int)(m.pk_per_transform * 4);
    // ONE bounds-checked resource over the whole image: a transform is an SGPR offset, a lane that must read zeros an
    // offset beyond the image
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)m.packed, 0, blk_bytes * T, 0x00020000);
    const int vo_lane = lane << 4;
    const int vo_T = chain_vo_T(lane);
    const int vo_q = q << 4;

    auto lds_bar = []() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // the tiles' degree words in LDS (a global load in the middle of a tile costs the chain an L2 round trip); two
    // words of padding behind them: "no groups" for the tile after the last
    // Per hidden tile, 16 words (filled once; two rows of "no groups" behind the last tile): the ranks its groups produce
    // (word 0 also carries the quad pattern << 16), and what depends on a rank alone of the addresses the chain needs --
    // byte offset of the rank's x / y word, of its (shift, raw) pair in a staged output tile, of its two rows in an
    // output fragment record (out of range for a padding group: the bounds-checked load returns zeros).
    int* DGT = reinterpret_cast<int*>(smem + TRI5_SET_FLOATS(Dp, Hp, MAXO));
    // between two transforms the chain wave starts the next sweep with rank 0; the constants it needs come from LDS tables
    // filled once -- B3T[t]: (shift, raw log-scale) of rank 0; PRM[r]: the feature of rank r of the last transform (whose x
    // is stored by feature).  x alternates between two arrays.
    int* PRM = DGT + TRI5_TT_WORDS(&m);
    float* B3T = reinterpret_cast<float*>(PRM + Dp);
    float* XB = reinterpret_cast<float*>(DGT + ((TRI5_TT_WORDS(&m) + Dp + 2 * T + 3) & ~3));      // (16-byte aligned)
    // No re-ranking between transforms: transform t reads its input y where the previous transform (t + 1) left it --
    // YT[t][tile][group]: byte offset (walker 0) of the y word of the rank the group produces, in the x array of transform
    // t + 1 (by ITS ranks), or in Y for the first transform; Y0T[t]: the same for rank 0.
    int* YT = reinterpret_cast<int*>(XB + Dp * 16);
    int* Y0T = YT + T * (nT + 2) * 4;
    auto fill_table = [&]() {
        for (int e = lane; e < (nT + 2) * 16; e += 64) {
            const int tile = e >> 4, k = e & 15, i = k & 3;
            int g = D, pat = 1;
            if (tile < nT) {
                int4 dg = *reinterpret_cast<const int4*>(quad_meta + 4 * tile);
                dg.x &= 0xffff; dg.y &= 0xffff; dg.z &= 0xffff; dg.w &= 0xffff;
                const bool ny = dg.y != dg.x, nz = dg.z != dg.y, nw = dg.w != dg.z;
                pat = 1 | (ny << 1) | (nz << 2) | (nw << 3);
                const int g1 = ny ? dg.y : (nz ? dg.z : (nw ? dg.w : D));
                const int g2 = ny ? (nz ? dg.z : (nw ? dg.w : D)) : ((nz && nw) ? dg.w : D);
                const int g3 = (ny && nz && nw) ? dg.w : D;
                g = i == 0 ? dg.x : (i == 1 ? g1 : (i == 2 ? g2 : g3));
            }
            const bool live = g < D;
            const int gg = live ? g : 0;
            int v;
            if (k < 4) v = g | (i == 0 ? pat << 16 : 0);
            else if (k < 8) v = 4 * (((gg >> 4) << 8) + ((gg & 3) << 6) + ((gg >> 2) & 3));
            else if (k < 12) v = 4 * ((gg >> 3) * 256 + 2 * (gg & 7));
            else v = live ? ((((gg >> 3) * nT) << 6) + 2 * (gg & 7)) << 4 : OOB_VOFF;
            DGT[e] = v;
        }
    };

    // the chain's operands of tile U of transform tt, request number K of 16 (maf_chain_rot.h: ChainFrags); gU / gV: the
    // ranks of tile U's / tile U+1's groups
    // the chain's operands of tile U of transform tt, request number K of 16 (maf_chain_rot.h: ChainFrags); wvU / wvV: the
    // fragment-row offsets of tile U's / tile U+1's groups (table words 12..15)
    auto request = [&](ChainFrags<MAXO>& F, auto k_, const int tt, const int U, const int4 wvU, const int4 wvV) {
        constexpr int K = decltype(k_)::value;
        const int base = tt * blk_bytes;
        const int Un = U + 1 < nT ? U + 1 : U;
        const int voN = U + 1 < nT ? vo_T : OOB_VOFF;
        // the A operand of the output MFMA: tile row i = lane & 15 carries, for i < 8, (shift, raw) of the tile's groups 0 and
        // 1 (i & 3 = 0, 1 / 2, 3), for i >= 8 those of groups 2 and 3 -- the wavefront's lower half receives the first pair
        // of groups, its upper half the second (maf_chain_rot.h: half mode)
        auto rows = [&](const int4 wv) {
            const int w_even = (lane & 8) ? wv.z : wv.x, w_odd = (lane & 8) ? wv.w : wv.y;
            return bload4(rs, ((lane & 2) ? w_odd : w_even) + (q << 8) + ((lane & 1) << 4), base + oF3 + U * 1024);
        };
        if constexpr (K == 0) F.wt1 = bload4(rs, vo_T, base + oF1 + (U * nT + U) * 1024);
        else if constexpr (K == 1) F.wt2 = bload4(rs, vo_T, base + oF2 + (U * nT + U) * 1024);
        else if constexpr (K == 2) F.wn1 = bload4(rs, voN, base + oF1 + (Un * nT + U) * 1024);
        else if constexpr (K == 3) F.wn2 = bload4(rs, voN, base + oF2 + (Un * nT + U) * 1024);
        else if constexpr (K == 4) F.wo[0] = rows(wvU);
        else if constexpr (K == 5) { }
        else if constexpr (K == 6) F.woN[0] = rows(wvV);
        else if constexpr (K == 7) { }
        else if constexpr (K < 11) F.w0o[K - 8] = bload4(rs, ((4 + K - 8) << 6) + vo_q, base + oCW0 + U * 1024);
        else if constexpr (K == 11) F.w0o[3] = make_float4(0.f, 0.f, 0.f, 0.f);      // (the fourth group has no later quad)
        else F.w0N[K - 12] = bload4(rs, U + 1 < nT ? ((K - 12) << 6) + vo_q : OOB_VOFF, base + oCW0 + Un * 1024);
    };
    // a tile's table words into its operand set
    auto take_table = [&](ChainFrags<MAXO>& F, const int tt, const int U) {
        const int4 ty = *reinterpret_cast<const int4*>(YT + (tt * (nT + 2) + U) * 4);
        F.yo[0] = ty.x; F.yo[1] = ty.y; F.yo[2] = ty.z; F.yo[3] = ty.w;
        const int4 tg = *reinterpret_cast<const int4*>(DGT + 16 * U);
        const int4 txy = *reinterpret_cast<const int4*>(DGT + 16 * U + 4);
        const int4 tso = *reinterpret_cast<const int4*>(DGT + 16 * U + 8);
        F.g[0] = tg.x & 0xffff; F.g[1] = tg.y; F.g[2] = tg.z; F.g[3] = tg.w;
        F.pat = tg.x >> 16;
        F.xy[0] = txy.x; F.xy[1] = txy.y; F.xy[2] = txy.z; F.xy[3] = txy.w;
        F.so[0] = tso.x; F.so[1] = tso.y; F.so[2] = tso.z; F.so[3] = tso.w;
    };

    ChainFrags<MAXO> fA, fB;
    if (wv < TRI5_NC) {
        // the first tile's operands are on their way while the walkers are proposed / loaded
        fill_table();
        WAVE_LDS_FENCE();
        {
            auto woff = [](const int r) { return 4 * (((r >> 4) << 8) + ((r & 3) << 6) + ((r >> 2) & 3)); };
            auto src_rank = [&](const int tt, const int g) {       // where rank g of transform tt sits in its input array
                if (g >= D) return 0;
                return tt == T - 1 ? g : rank_of_feat[(tt + 1) * D + feat_of_rank[tt * D + g]];
            };
            for (int e = lane; e < T * (nT + 2) * 4; e += 64) {
                const int tt = e / ((nT + 2) * 4), rem = e - tt * (nT + 2) * 4;
                YT[e] = woff(src_rank(tt, DGT[16 * (rem >> 2) + (rem & 3)] & 0xffff));
            }
            for (int tt = lane; tt < T; tt += 64) Y0T[tt] = woff(src_rank(tt, 0));
        }
        WAVE_LDS_FENCE();
        {
            const int4 w0 = *reinterpret_cast<const int4*>(DGT + 12), w1 = *reinterpret_cast<const int4*>(DGT + 16 + 12);
            take_table(fA, T - 1, 0);
            static_for<16>([&](auto k_) { request(fA, k_, T - 1, 0, w0, w1); });
        }
        if constexpr (FM > 0) {
            for (int e = lane; e < (Dp - D) * 16; e += 64) Y[lidx(D + (e >> 4), e & 15)] = 0.0f;
            const double sg = pa.adapt ? pa.adapt[0] : pa.sigma, ca = pa.adapt ? pa.adapt[1] : pa.cn_a;
            propose_body<FM>(pa.kind, pa.cur32, nullptr, pa.adapt ? pa.adapt + 2 : pa.mu, pa.inv_cov, pa.chol, pa.nu, sg,
                             ca, pa.rng, pa.prop64, nullptr, pa.quad, pa.quad_prop, n, D, Y, rank_of_feat + (T - 1) * D,
                             set);
        } else {
            load_rows(Y, in, row0, n, D, Dp, feat_of_rank + (T - 1) * D, lane);
        }
    } else {
        // padding slots of the activations are read by the bursts (times zero weights): zero once
        float4* z4 = reinterpret_cast<float4*>(H0);
        const int n4 = (2 * Hp * 16 + 2 * 256 + 2 * TRI5_STAGE_FLOATS(MAXO)) >> 2;
        for (int e = lane; e < n4; e += 64) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (pa.prof && lane == 0 && blockIdx.x < 64)         // (measurement only: which SIMD / CU every wavefront landed on)
        pa.prof[(size_t)T * nT * 8 + blockIdx.x * 2 + wv] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    for (int r = threadIdx.x; r < D; r += 64 * (TRI5_NC + 1)) PRM[r] = feat_of_rank[r];      // (the last transform's x is stored by feature)
    for (int tt = threadIdx.x; tt < T; tt += 64 * (TRI5_NC + 1)) {
        const float* b3 = m.packed + (size_t)tt * m.pk_per_transform + (oB3 >> 2);
        B3T[2 * tt] = b3[0];
        B3T[2 * tt + 1] = b3[1];
    }
    for (int e = threadIdx.x; e < (Dp * 16) >> 2; e += 64 * (TRI5_NC + 1)) {      // both x arrays start zeroed
        reinterpret_cast<float4*>(XA)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4*>(XB)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float ladj = 0.0f;
    int xsel = 0;
    __syncthreads();

    // everything the preparation of tile TT needs from global memory: the hidden layers' fragments against tiles
    // 0 .. TT-2 (transposed rows), the masked layer-0 fragments, the transposed biases, the output fragments
            // against tile TT-2
#define BURST_FETCH(TB, TT, P1, P2, XF, OF, Bz0, Bz1, Bz2)                                                        \
            {   /* unconditional: a request under a branch makes the register set a conditional assignment -- copies */ \
                /* behind a wait for the loads just issued; what a tile does not need is requested out of range (zeros) */ \
                const int TT_ = (TT) < nT ? (TT) : nT - 1;                                                        \
                const int so1_ = (TB) + oF1 + TT_ * nT * 1024, so2_ = (TB) + oF2 + TT_ * nT * 1024;               \
                _Pragma("unroll") for (int i_ = 0; i_ < PK4; ++i_) {                                              \
                    const int vo_ = i_ < TT_ - 1 ? vo_T : OOB_VOFF;                                               \
                    P1[i_] = bload4(rs, vo_, so1_ + i_ * 1024);                                                   \
                    P2[i_] = bload4(rs, vo_, so2_ + i_ * 1024);                                                   \
                }                                                                                                 \
                _Pragma("unroll") for (int i_ = 0; i_ < PXB; ++i_)                                                \
                    XF[i_] = bload4(rs, i_ < nXT ? vo_T : OOB_VOFF, (TB) + oF0C + (TT_ * nXT + i_) * 1024);        \
                Bz0 = bload4(rs, vo_q, (TB) + oB0T + 64 * TT_);                                                   \
                Bz1 = bload4(rs, vo_q, (TB) + oB1T + 64 * TT_);                                                   \
                Bz2 = bload4(rs, vo_q, (TB) + oB2T + 64 * TT_);                                                   \
                _Pragma("unroll") for (int O = 0; O < MAXO; ++O)                                                  \
                    OF[O] = bload4(rs, (TT_ >= 2 && O < nOT) ? vo_lane : OOB_VOFF, (TB) + oF3 + (O * nT + (TT_ >= 2 ? TT_ - 2 : 0)) * 1024); \
            }
    // The two wavefronts run loops of their own (their operand sets would otherwise be live across each other's code);
    // they meet at the barriers: nTl + 1 LDS-only ones and one full one per transform.
    if (wv == TRI5_NC) {
        // Two operand sets, used alternately: while tile T1 is being prepared from one, the operands of tile T1 + 1
        // are already on their way into the other (nothing else hides their L2 latency here).
        float4 pA1[PK4], pA2[PK4], pB1[PK4], pB2[PK4], xA[PXB], xB[PXB], oA[MAXO], oB[MAXO], ob[MAXO];
        float4 bA0, bA1, bA2, bB0, bB1, bB2;
        // The first tile of a transform needs nothing of the transform (biases only): it is staged while the chain still
        // runs the LAST tile of the transform before (the staging buffers alternate across the boundary: `spar`), so that
        // at a transform boundary the chain solves rank 0 and goes on.
        int tb = (T - 1) * blk_bytes, spar = 0;
        // (what the x array holds on entry -- zeros, or the x of two transforms ago -- meets zero weights only: the layer-0
        //  fragments f0c carry the columns of the ranks that are final; the other array is the chain's y)
        float* X = XA;
        f32x4 oN[MAXO];                                    // output-layer partials of the walker set (right-looking, natural layout)
        for (int t = T - 1; t >= 0; --t) {
            // ------------------------------------------------------------------ BURST wave
#define BURST_K(NK, P1, P2, AA1, AA2)                                                                             \
            switch ((NK) < PK4 ? (NK) : PK4) {                                                                    \
                case 1: burst_tile_chain<1>(AA1, AA2, P1, P2, H0, H1, lane); break;                               \
                case 2: burst_tile_chain<2>(AA1, AA2, P1, P2, H0, H1, lane); break;                               \
                case 3: burst_tile_chain<3>(AA1, AA2, P1, P2, H0, H1, lane); break;                               \
                case 4: burst_tile_chain<4>(AA1, AA2, P1, P2, H0, H1, lane); break;                               \
                case 5: burst_tile_chain<5>(AA1, AA2, P1, P2, H0, H1, lane); break;                               \
                case 6: burst_tile_chain<6>(AA1, AA2, P1, P2, H0, H1, lane); break;                               \
                case 7: burst_tile_chain<7>(AA1, AA2, P1, P2, H0, H1, lane); break;                               \
                case 8: burst_tile_chain<8>(AA1, AA2, P1, P2, H0, H1, lane); break;                               \
                default: break;                                                                                   \
            }
            // prepare tile TT from set (P1 ...) while the chain runs tile TT-1; request tile TT+1 into set (N1 ...)
#define BURST_TILE(TT, P1, P2, XF, OF, Bz0, Bz1, Bz2, N1, N2, NXF, NOF, Nz0, Nz1, Nz2)                             \
            {                                                                                                     \
                const int T1 = (TT);                                                                              \
                BURST_FETCH(tb, T1 + 1, N1, N2, NXF, NOF, Nz0, Nz1, Nz2)                                          \
                f32x4 a0, a1, a2;                                                                                 \
                a0[0] = Bz0.x; a0[1] = Bz0.y; a0[2] = Bz0.z; a0[3] = Bz0.w;                                       \
                a1[0] = Bz1.x; a1[1] = Bz1.y; a1[2] = Bz1.z; a1[3] = Bz1.w;                                       \
                a2[0] = Bz2.x; a2[1] = Bz2.y; a2[2] = Bz2.z; a2[3] = Bz2.w;                                       \
                const int nK = T1 - 1;                          /* hidden tiles 0 .. T1-2 are final */            \
                BURST_K(nK, P1, P2, a1, a2)                                                             \
                if (nK > PK4) {                                 /* flows wider than PK4 + 2 tiles: four K tiles' fragments in flight */ \
                    float4 w1r[4], w2r[4];                                                                        \
                    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                            \
                        const int Kl = min(PK4 + j_, nT - 1);                                                     \
                        w1r[j_] = bload4(rs, vo_T, tb + oF1 + (T1 * nT + Kl) * 1024);                             \
                        w2r[j_] = bload4(rs, vo_T, tb + oF2 + (T1 * nT + Kl) * 1024);                             \
                    }                                                                                             \
                    for (int K0 = PK4; K0 < nK; K0 += 4) {                                                        \
                        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                        \
                            const int K = K0 + j_;                                                                \
                            const float4 w1 = w1r[j_], w2 = w2r[j_];                                              \
                            const int Kn = min(K + 4, nT - 1);                                                    \
                            w1r[j_] = bload4(rs, vo_T, tb + oF1 + (T1 * nT + Kn) * 1024);                         \
                            w2r[j_] = bload4(rs, vo_T, tb + oF2 + (T1 * nT + Kn) * 1024);                         \
                            if (K < nK) {                                                                         \
                                const float4 b1 = *reinterpret_cast<const float4*>(H0 + (K << 8) + (lane << 2));  \
                                const float4 b2 = *reinterpret_cast<const float4*>(H1 + (K << 8) + (lane << 2));  \
                                a1 = MFMA(w1.x, b1.x, a1); a2 = MFMA(w2.x, b2.x, a2);                             \
                                a1 = MFMA(w1.y, b1.y, a1); a2 = MFMA(w2.y, b2.y, a2);                             \
                                a1 = MFMA(w1.z, b1.z, a1); a2 = MFMA(w2.z, b2.z, a2);                             \
                                a1 = MFMA(w1.w, b1.w, a1); a2 = MFMA(w2.w, b2.w, a2);                             \
                            }                                                                                     \
                        }                                                                                         \
                    }                                                                                             \
                }                                                                                                 \
                /* layer 0 against the ranks of tiles <= T1-2 (the chain adds the ranks of tile T1-1 itself) */   \
                _Pragma("unroll") for (int i_ = 0; i_ < PXB; ++i_) {                                              \
                    if (i_ < nXT) {                                                                               \
                        const float4 b = *reinterpret_cast<const float4*>(X + (i_ << 8) + (lane << 2));           \
                        a0 = MFMA(XF[i_].x, b.x, a0); a0 = MFMA(XF[i_].y, b.y, a0);                               \
                        a0 = MFMA(XF[i_].z, b.z, a0); a0 = MFMA(XF[i_].w, b.w, a0);                               \
                    }                                                                                             \
                }                                                                                                 \
                /* output rows: tile T1-2's h2 into every output tile that has a rank above its degrees */        \
                if (T1 >= 2) {                                                                                    \
                    const int O0 = (__builtin_amdgcn_readfirstlane(DGT[16 * (T1 - 2)]) & 0xffff) >> 3;                                             \
                    const float4 b = *reinterpret_cast<const float4*>(H2 + (((T1 - 2) & 1) << 8) + (lane << 2));  \
                    _Pragma("unroll") for (int O = 0; O < MAXO; ++O) {                                            \
                        if (O >= O0 && O < nOT) {                                                                 \
                            oN[O] = MFMA(OF[O].x, b.x, oN[O]); oN[O] = MFMA(OF[O].y, b.y, oN[O]);                 \
                            oN[O] = MFMA(OF[O].z, b.z, oN[O]); oN[O] = MFMA(OF[O].w, b.w, oN[O]);                 \
                        }                                                                                         \
                    }                                                                                             \
                }                                                                                                 \
                float* st_ = STG + ((T1 + spar) & 1) * TRI5_STAGE_FLOATS(MAXO);                                   \
                *reinterpret_cast<float4*>(st_ + (lane << 2)) = make_float4(a0[0], a0[1], a0[2], a0[3]);          \
                *reinterpret_cast<float4*>(st_ + 256 + (lane << 2)) = make_float4(a1[0], a1[1], a1[2], a1[3]);    \
                *reinterpret_cast<float4*>(st_ + 512 + (lane << 2)) = make_float4(a2[0], a2[1], a2[2], a2[3]);    \
                _Pragma("unroll") for (int O = 0; O < MAXO; ++O)                                                  \
                    *reinterpret_cast<float4*>(st_ + 768 + O * 256 + (p << 4) + (q << 2)) = make_float4(oN[O][0], oN[O][1], oN[O][2], oN[O][3]); \
                if (T1 > 0) lds_bar();                                    /* E(T1 - 1) */                         \
            }
            // the first tile of transform tt (block offset tb, x array X, staging parity spar: set by the caller)
#define BURST_FIRST()                                                                                             \
            {                                                                                                     \
                BURST_FETCH(tb, 0, pA1, pA2, xA, oA, bA0, bA1, bA2)                                               \
                _Pragma("unroll") for (int O = 0; O < MAXO; ++O) ob[O] = bload4(rs, O < nOT ? vo_q : OOB_VOFF, tb + oB3 + 64 * O); \
                _Pragma("unroll") for (int O = 0; O < MAXO; ++O) { oN[O][0] = ob[O].x; oN[O][1] = ob[O].y; oN[O][2] = ob[O].z; oN[O][3] = ob[O].w; } \
                BURST_TILE(0, pA1, pA2, xA, oA, bA0, bA1, bA2, pB1, pB2, xB, oB, bB0, bB1, bB2)                    \
            }
            if (t == T - 1) BURST_FIRST()
            lds_bar();                                                    // E(-1): the chain solved rank 0
            for (int T2 = 1; T2 < nTl; T2 += 2) {
                BURST_TILE(T2, pB1, pB2, xB, oB, bB0, bB1, bB2, pA1, pA2, xA, oA, bA0, bA1, bA2)
                if (T2 + 1 >= nTl) break;
                BURST_TILE(T2 + 1, pA1, pA2, xA, oA, bA0, bA1, bA2, pB1, pB2, xB, oB, bB0, bB1, bB2)
            }
            if (t > 0) {                                                  // the next transform's first tile, while the chain runs this one's last
                tb = (t - 1) * blk_bytes;
                X = (X == XA) ? XB : XA;
                spar = (spar + nTl) & 1;
                BURST_FIRST()
            }
#undef BURST_FIRST
#undef BURST_TILE
#undef BURST_K
#undef BURST_FETCH
            lds_bar();                                                    // E(nTl - 1)
            if (t == 0) { __syncthreads(); xsel = (X == XB); }            // (the chain stored the result; xsel names the last x array)
        }
    } else {
        float4 w00 = bload4(rs, vo_q, (T - 1) * blk_bytes + oCW0);      // layer 0, first tile: the column of rank 0
        const float* Ysrc = Y;                             // the input of the transform: Y, then the previous transform's x array
        int spar = 0;                                      // staging parity of the transform's first tile (the buffers alternate across transforms)
        for (int t = T - 1; t >= 0; --t) {
            float* X = xsel ? XB : XA;
            xsel ^= 1;
            // ------------------------------------------------------------------ CHAIN wave
            ChainRot<MAXO> s;
            {   // rank 0 reads nothing: bias only; its share of the first tile's layer 0 (the window's first slot)
                const float shift = B3T[2 * t], ls = fast_ls(B3T[2 * t + 1]);
                const float y0 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(Ysrc) + (p << 4) + Y0T[t]);
                const float xv = (y0 - shift) * fast_exp_neg(ls);
                ladj -= ls;
                if (q == 0) X[lidx(0, p)] = xv;
                s.a0N[0] = w00.x * xv; s.a0N[1] = w00.y * xv; s.a0N[2] = w00.z * xv; s.a0N[3] = w00.w * xv;
            }
            s.accN1 = s.accN2 = s.outN[0] = s.outN[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            lds_bar();                                                    // E(-1): the first tile's staging is complete

            // one tile from the operands in `cur`, the next tile's requested into `nxt`
            auto tile = [&](ChainFrags<MAXO>& cur, ChainFrags<MAXO>& nxt, const int Tt_, auto fast_) {
                constexpr bool FAST = decltype(fast_)::value;          // the caller knows the tile has four single-quad groups
                const int Tt = __builtin_amdgcn_readfirstlane(Tt_);
                long long* pf = (pa.prof && blockIdx.x == 0) ? pa.prof + ((size_t)(T - 1 - t) * nT + Tt) * 8 : nullptr;
                if (pf && lane == 0) pf[0] = clock64();
                float* st = STG + ((Tt + spar) & 1) * TRI5_STAGE_FLOATS(MAXO);
                const float4 s0 = *reinterpret_cast<const float4*>(st + (lane << 2));
                const float4 s1 = *reinterpret_cast<const float4*>(st + 256 + (lane << 2));
                const float4 s2 = *reinterpret_cast<const float4*>(st + 512 + (lane << 2));
                const int pat = __builtin_amdgcn_readfirstlane(cur.pat);
                const char* stb = reinterpret_cast<const char*>(st) + 3072 + (p << 6);      // the walker's row of the staged output tiles
                // (half mode: a lane keeps the output partials and the y of ITS pair of groups -- 0, 1 for q < 2; 2, 3 above)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int so_k = q < 2 ? cur.so[k] : cur.so[2 + k], yo_k = q < 2 ? cur.yo[k] : cur.yo[2 + k];
                    const float2 so = *reinterpret_cast<const float2*>(stb + so_k);
                    s.po[k] = make_float2(so.x + s.outN[0][2 * k], so.y + s.outN[0][2 * k + 1]);
                    s.yv[k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(Ysrc) + (p << 4) + yo_k);
                }
                s.a0[0] = s0.x + s.a0N[0]; s.a0[1] = s0.y + s.a0N[1]; s.a0[2] = s0.z + s.a0N[2]; s.a0[3] = s0.w + s.a0N[3];
                s.p1[0] = s1.x + s.accN1[0]; s.p1[1] = s1.y + s.accN1[1]; s.p1[2] = s1.z + s.accN1[2]; s.p1[3] = s1.w + s.accN1[3];
                s.p2[0] = s2.x + s.accN2[0]; s.p2[1] = s2.y + s.accN2[1]; s.p2[2] = s2.z + s.accN2[2]; s.p2[3] = s2.w + s.accN2[3];
#pragma unroll
                for (int j = 0; j < 4; ++j) s.a0N[j] = 0.0f;
                s.accN1 = s.accN2 = s.outN[0] = s.outN[1] = f32x4{0.f, 0.f, 0.f, 0.f};
                s.acc1 = s.acc2 = s.outR[0] = s.outR[1] = f32x4{0.f, 0.f, 0.f, 0.f};
                {   // where the groups' x go (the lanes that do not own the word: a scratch word of their own, the staging buffer just read)
                    float* scratch = st + lane;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        s.xa[i] = (q == 0 && cur.g[i] < D) ? reinterpret_cast<float*>(reinterpret_cast<char*>(X) + (p << 4) + cur.xy[i]) : scratch;
                    s.pend_a = scratch; s.pend_x = 0.0f; s.pend_ls = 0.0f;
                }
                // the next tile's operands (the first tile of the next transform behind the last): requested in the
                // shadows of this tile's hops, 16 requests over 3 x (groups) shadows
                // (every tile requests: behind the last tile of the last transform the loads are harmless and unused --
                //  a conditional request would make the register set a conditional assignment, i.e. copies)
                const bool more = Tt + 1 < nTl;
                const int ntt = more ? t : (t > 0 ? t - 1 : 0), nU = more ? Tt + 1 : 0;
                take_table(nxt, ntt, nU);
                const int4 gU = *reinterpret_cast<const int4*>(DGT + 16 * nU + 12);
                const int4 gV = *reinterpret_cast<const int4*>(DGT + 16 * (nU + 1) + 12);
                auto ahead = [&](auto gi_, auto hop_, auto ng_) {
                    constexpr int G = decltype(gi_)::value, HP = decltype(hop_)::value, NG_ = decltype(ng_)::value, NSH = 3 * NG_;
                    constexpr int LPS = (16 + NSH - 1) / NSH, k0 = (G * 3 + HP) * LPS;
                    auto into = [&](ChainFrags<MAXO>& F, auto k_) { request(F, k_, ntt, nU, gU, gV); };
                    using std::integral_constant;
                    if constexpr ((TRI5_ABL & 0x100) != 0) {
                    } else if constexpr (FAST && NG_ == 4) {
                        // the common tile: the layer-0 window columns of a group die with the group, so the next tile's go
                        // straight into the CURRENT set once it has run; only what lives to the tile's end is buffered
                        // (and copied over at the boundary: 7 of 13 operands)
                        constexpr int sh = 3 * G + HP;
                        if constexpr (sh == 0) { into(nxt, integral_constant<int, 0>{}); into(nxt, integral_constant<int, 1>{}); }
                        else if constexpr (sh == 1) { into(nxt, integral_constant<int, 2>{}); into(nxt, integral_constant<int, 3>{}); }
                        else if constexpr (sh == 2) { into(nxt, integral_constant<int, 4>{}); into(nxt, integral_constant<int, 6>{}); }
                        else if constexpr (sh == 3) { into(cur, integral_constant<int, 8>{}); into(cur, integral_constant<int, 12>{}); }
                        else if constexpr (sh == 4) { into(nxt, integral_constant<int, 15>{}); }
                        else if constexpr (sh == 6) { into(cur, integral_constant<int, 9>{}); into(cur, integral_constant<int, 13>{}); }
                        else if constexpr (sh == 9) { into(cur, integral_constant<int, 10>{}); into(cur, integral_constant<int, 14>{}); }
                    } else {
                        static_for<LPS>([&](auto j_) {
                            constexpr int K = k0 + decltype(j_)::value;
                            if constexpr (K < 16) into(nxt, integral_constant<int, K>{});
                        });
                    }
                };
                if (pf && lane == 0) pf[1] = clock64();
                if constexpr (FAST || (TRI5_ABL & 0x200) != 0) {   // four single-quad groups: the common tile
                    chain_group_rot<15, 0, 4, MAXO, TRI5_ABL | 11>(s, cur, H0, H1, X, Tt, D, nOT, q, p, ladj, H2, ahead);
                } else
                switch (pat) {
#define CASE(P) case P: chain_group_rot<P, 0, 4, MAXO, TRI5_ABL | 11>(s, cur, H0, H1, X, Tt, D, nOT, q, p, ladj, H2, ahead); break;
                    CASE(1) CASE(3) CASE(5) CASE(7) CASE(9) CASE(11) CASE(13) CASE(15)
#undef CASE
                    default: break;
                }
                chain_flush(s, ladj);
                if (pf && lane == 0) pf[2] = clock64();
                lds_bar();                                            // E(Tt): this tile is final
                if (pf && lane == 0) pf[3] = clock64();
            };
            // One tile body; the next tile's operands arrive in the second set and are moved over at the boundary (two bodies
            // with the sets swapping roles were slower: twice the code, and the second copy allocated worse).
            // Runs of common tiles are a loop of their own: a tile body that joins the other patterns' bodies pays for it
            // with ~45 register copies per join (every register a body updates becomes a conditional assignment), per tile.
            for (int Tt = 0; Tt < nTl;) {
                while (Tt < nTl && __builtin_amdgcn_readfirstlane(fA.pat) == 15 && !(TRI5_ABL & 0x100)) {
                    tile(fA, fB, Tt, std::true_type{});
                    // (the window columns were requested in place)
                    fA.wt1 = fB.wt1; fA.wt2 = fB.wt2; fA.wn1 = fB.wn1; fA.wn2 = fB.wn2;
                    fA.wo[0] = fB.wo[0]; fA.woN[0] = fB.woN[0]; fA.w0N[3] = fB.w0N[3];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { fA.g[i] = fB.g[i]; fA.xy[i] = fB.xy[i]; fA.so[i] = fB.so[i]; fA.yo[i] = fB.yo[i]; }
                    fA.pat = fB.pat;
                    ++Tt;
                }
                if (Tt >= nTl) break;
                tile(fA, fB, Tt, std::false_type{});
                fA = fB;
                ++Tt;
            }
            w00 = bload4(rs, vo_q, (t > 0 ? t - 1 : 0) * blk_bytes + oCW0);
            Ysrc = X;                                      // the next transform reads its y from here, through its offset table
            spar = (spar + nTl) & 1;
            if (t == 0) {
                const int* prm = PRM;                      // rank -> feature of the last transform inverted
                for (int e = lane; e < D * 16; e += 64) {
                    const int r = e >> 4, pp = e & 15;
                    if (row0 + pp < n) out[(row0 + pp) * D + prm[r]] = X[lidx(r, pp)];
                }
                __syncthreads();
                xsel ^= 1;                                 // (xsel names the array of the LAST transform again: the epilogue reads it)
            }
        }
    }
    float* X = xsel ? XB : XA;
    if (wv < TRI5_NC && ladj_out && lane < 16 && row0 + p < n) ladj_out[row0 + p] = ladj;
    if constexpr (FM > 0 && TRI5_NC == 1) {
        if (pa.epi.on)                                 // both wavefronts: 16 walkers x D elements over 128 threads
            scaler_epilogue(pa.epi, X, rank_of_feat, reinterpret_cast<double*>(H0), row0, n, D, (int)threadIdx.x,
                            64 * (TRI5_NC + 1), [](int r, int pp) { return lidx(r, pp); });
    }
}

// -1: automatic (by size), 0: never, 1: always
static int tri5_mode() {
    static const int mode = pmc_env_int("PMC_INVERSE_DUO", -1);
    return mode;
}

static bool tri5_wanted(const pmc_maf_t* m, int64_t n) {
    const int mode = tri5_mode();
    if (mode >= 0) return mode != 0;
    // The right-looking two-wave sweep takes ~0.55 of the lone wave's time per round, and a launch beyond the 512
    // resident walker sets simply runs its surplus workgroups as they find a CU: it is taken whenever its LDS fits.
    (void)n;
    const int maxo = m->nOT <= 4 ? 4 : 8;
    return (size_t)TRI5_LDS_FLOATS(m, maxo) * sizeof(float) <= 160 * 1024;
}

// which of the two D <= 64 sweeps PMC_INVERSE_AUTO launches for n rows (bench.py names the kernel it times with it)
extern "C" int pmc_maf_inverse_auto_is_duo(const pmc_maf_t* m, int64_t n) {
    if (!m || m->n_out != 2 || !m->tri_ok || m->nOT > 8 || m->D > 64) return 0;
    return tri5_wanted(m, n) ? 1 : 0;
}

// same contract as pmc_launch_propose_inverse_tri4 / pmc_launch_inverse_tri4 (pa == nullptr: plain inverse of z)
static int launch_tri5(const ProposeArgs* pa, const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n,
                       hipStream_t stream) {
    if (m->n_out != 2 || !m->tri_ok || m->nOT > 8 || m->D > 64) return -1;
    if (m->pk_per_transform * 4 * m->T >= (int64_t)OOB_VOFF) return -1;       // one buffer resource over the whole image
    const int maxo = m->nOT <= 4 ? 4 : 8;
    const size_t lds = (size_t)TRI5_LDS_FLOATS(m, maxo) * sizeof(float);
    if (lds > 160 * 1024) return -1;
    const ProposeArgs none{};
    const int64_t nsets = (n + 15) / 16;
    const unsigned grid = (unsigned)((nsets + TRI5_NC - 1) / TRI5_NC);
#define LAUNCH5(MO, FMV)                                                                                          \
    {                                                                                                             \
        if (lds > 48 * 1024) {                                                                                    \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(maf_inverse_tri5_kernel<MO, FMV>),   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
            if (e != hipSuccess) return pmc_fail_hip(e, "hipFuncSetAttribute(maf_inverse_tri5_kernel)");          \
        }                                                                                                         \
        hipLaunchKernelGGL((maf_inverse_tri5_kernel<MO, FMV>), dim3(grid), dim3(64 * (TRI5_NC + 1)), lds,          \
                           stream, *m, z, x, ladj, n, pa ? *pa : none);                                           \
    }
    if (!pa || !pa->cur32) { if (maxo == 4) LAUNCH5(4, 0) else LAUNCH5(8, 0) }       // (pa without a walker state: the profile entry)
    else if (m->D <= 16) { if (maxo == 4) LAUNCH5(4, 4) else LAUNCH5(8, 4) }
    else if (m->D <= 32) { if (maxo == 4) LAUNCH5(4, 8) else LAUNCH5(8, 8) }
    else { if (maxo == 4) LAUNCH5(4, 16) else LAUNCH5(8, 16) }
#undef LAUNCH5
    return pmc_check_launch("maf_inverse_tri5_kernel");
}

#ifdef PMC_DEBUG_HOOKS
// measurement only (scripts/profile_tri5.py): cycle stamps of the chain wave of workgroup 0 -- prof[transform * nT + tile][8]
extern "C" int pmc_debug_tri5_profile(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n, long long* prof,
                                      void* stream) {
    ProposeArgs pa{};
    pa.prof = prof;
    return launch_tri5(&pa, m, z, x, ladj, n, (hipStream_t)stream) < 0 ? pmc_fail("pmc_debug_tri5_profile: flow not covered") : 0;
}
#endif

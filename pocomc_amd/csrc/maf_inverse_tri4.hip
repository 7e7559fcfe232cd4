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

#define DG_WORDS(m) ((m)->nT * 4 + 4)      // LDS words of the tiles' degree table (4 per tile, padded)
// + the two-wave sweep's permutation / rank-0 tables and its second x array (with alignment slack)
#define TRI5_TABLE_WORDS(m) (((DG_WORDS(m) + (m)->T * (m)->Dp + 2 * (m)->T + 3) & ~3) + (m)->Dp * 16)
#include "propose_body.h"

#define PX4 2
#define PK4 8

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}


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
                s.g[0] = dg.x;
                s.g[1] = ny ? dg.y : (nz ? dg.z : (nw ? dg.w : D));
                s.g[2] = ny ? (nz ? dg.z : (nw ? dg.w : D)) : ((nz && nw) ? dg.w : D);
                s.g[3] = (ny && nz && nw) ? dg.w : D;
            }
            const int soD1 = oF1 + (Tt * nT + Tt) * 1024, soD2 = oF2 + (Tt * nT + Tt) * 1024;
            if (!(ABL & 32)) {
            s.wt1 = bload4(rs, vo_T, soD1);
            s.wt2 = bload4(rs, vo_T, soD2);
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                // rows (lane&3): 0,1 -> (shift, raw) of group 2*sl; 2,3 -> group 2*sl+1
                const int g_even = s.g[2 * sl], g_odd = s.g[2 * sl + 1];
                const int gsel = (lane & 2) ? g_odd : g_even;
                const bool ok = gsel < D;
                const int gg = ok ? gsel : 0;
                const int vo = ((((gg >> 3) * nT) << 6) + (q << 4) + 2 * (gg & 7) + (lane & 1)) << 4;
                const float4 v = bload4(rs, vo, oF3 + Tt * 1024);
                s.wo[sl] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int O = 0; O < MAXO; ++O)
                s.f3n[O] = (O < nOT) ? bload4(rs, vo_lane, oF3 + (O * nT + Tt) * 1024) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int gg = s.g[i] < D ? s.g[i] : 0;
#pragma unroll
                for (int jt = i + 1; jt < 4; ++jt)
                    s.w0r[i][jt] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        rs, q << 2, oW0 + (gg * Hp + 16 * Tt + 4 * jt) * 4, 0));
            }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) s.yv[i] = Y[lidx(s.g[i] < D ? s.g[i] : 0, p)];

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
                const int gg = s.g[i] < D ? s.g[i] : 0;
                s.po[i] = *reinterpret_cast<const float2*>(SO + (gg >> 3) * 256 + (p << 4) + 2 * (gg & 7));
            }

            // first group, then the next tile's burst fragments (loads return in order: issued any earlier
            // they would sit between the chain and its own fragments), then the remaining groups
            chain_tile_begin(s, X, S, D, q, p, lane);
            if (!(ABL & 8))
            switch (pat) {
#define CASE(P) case P: chain_group_rot<P, 0, 1, MAXO, ABL>(s, H0, H1, X, Tt, D, nOT, q, p, ladj); break;
                CASE(1) CASE(3) CASE(5) CASE(7) CASE(9) CASE(11) CASE(13) CASE(15)
#undef CASE
            }
            if (Tt + 1 < nT) {
                PREFETCH4(Tt + 1);
                dg_next = *reinterpret_cast<const int4*>(DGT + 4 * (Tt + 1));
            }
            if (!(ABL & 8))
            switch (pat) {
#define CASE(P) case P: chain_group_rot<P, 1, 4, MAXO, ABL>(s, H0, H1, X, Tt, D, nOT, q, p, ladj); break;
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
    static const bool on = getenv("PMC_INVERSE_LANE") && atoi(getenv("PMC_INVERSE_LANE")) != 0;
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
// tri5: the same sweep with a BURST wave next to the chain waves.
//
// Ablations of tri4 (scripts/ablate_inverse.py): of its 100 us, 29 us are the left-looking bursts of the hidden
// layers and 14 us the issue of their fragment prefetches -- work that does not depend on the chain of the tile
// that is running.  Here a workgroup is three wavefronts: waves 0 and 1 (CHAIN waves) own 16 walkers each and keep
// the dependent work -- the layer-0 burst against x, the per-group chain, the right-looking output updates -- and
// wave 2 (the BURST wave) prepares, one tile ahead and for both walker sets from ONE set of weight fragments, the
// layer-1/2 pre-activations of the next tile: everything against the tiles that are already final while the
// chains run, the last K tile right after them.  Two LDS-only barriers per tile: A(t) "the chains of tile t
// finished" and B(t) "the pre-activations of tile t are in the staging areas".
// Where it pays: a workgroup lives on one CU (4 SIMDs, one such 256-VGPR wave each), so TRI5_NC = 1 runs at most
// 512 walker sets at a time and TRI5_NC = 2 (one burst wave for two sets: 3 waves) 256 workgroups = 512 sets too.
// Up to 8192 walkers the sweep takes ~68 us instead of tri4's ~90 us; 1e4 walkers (625 sets) need a second round
// (140 us) where tri4's 625 single waves are all resident (100 us).  The launcher therefore picks this kernel for
// n <= 16 * (sets resident at once) -- pocoMC's own default, n_active = 256, is deep inside that range -- and tri4
// above it.  PMC_INVERSE_DUO=0 / 1 forces never / always (tests run both).
// ============================================================================================================
#define TRI5_NC 1                  // chain waves (16-walker sets) per workgroup; see the note on occupancy above
#ifndef TRI5_ABL
#define TRI5_ABL 0                 // timing experiments only (maf_chain_rot.h): results are wrong when != 0
#endif

template <int MAXO, int FM>
__global__ __launch_bounds__(64 * (TRI5_NC + 1)) void maf_inverse_tri5_kernel(pmc_maf_t m, const float* __restrict__ in,
                                                                              float* __restrict__ out,
                                                                              float* __restrict__ ladj_out, int64_t n,
                                                                              ProposeArgs pa) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, p = lane & 15;
    const int D = m.D, Dp = m.Dp, Hp = m.Hp, T = m.T, nT = m.nT, nXT = m.nXT, nOT = m.nOT;
    const int set_floats = 2 * Dp * 16 + 2 * Hp * 16 + 2 * 256 + 3 * 256 + MAXO * 256;   // LDS of one walker set
    const int cs = wv < TRI5_NC ? wv : 0;                                        // this chain wave's set in the workgroup
    const int64_t set = (int64_t)blockIdx.x * TRI5_NC + cs;
    const int64_t row0 = set * 16;
    float* Y = smem + (size_t)cs * set_floats;
    float* XA = Y + Dp * 16;
    float* H0 = XA + Dp * 16;
    float* H1 = H0 + Hp * 16;
    float* H2 = H1 + Hp * 16;                  // h2 of the last two tiles [tile parity][256]: the burst wave's output updates read it
    float* S = H2 + 2 * 256;                   // staging: [3 layers][16 p][16 rows] then [MAXO][16 p][16 rows]
    float* SO = S + 3 * 256;
    const int* feat_of_rank = m.meta + 8;
    const int* rank_of_feat = m.meta + 8 + T * D;
    const int* quad_meta = m.meta + 8 + 2 * T * D;

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
    const int vo_lane = lane << 4;
    const int vo_T = chain_vo_T(lane);
    const int vo_q = q << 4;

    auto lds_bar = []() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    if (wv < TRI5_NC) {
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
        for (int c = 0; c < TRI5_NC; ++c) {    // padding slots of the activations are read by the bursts: zero once
            float4* z4 = reinterpret_cast<float4*>(smem + (size_t)c * set_floats + 2 * Dp * 16);
            const int n4 = (2 * Hp * 16 + 2 * 256) >> 2;
            for (int e = lane; e < n4; e += 64) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (pa.prof && lane == 0 && blockIdx.x < 64)         // (measurement only: which SIMD / CU every wavefront landed on)
        pa.prof[(size_t)T * nT * 8 + blockIdx.x * 2 + wv] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    // the tiles' degree words in LDS (see maf_inverse_tri4_kernel); filled by the burst wave, visible after the first
    // __syncthreads() of the transform loop
    int* DGT = reinterpret_cast<int*>(smem + (size_t)TRI5_NC * set_floats);
    if (wv == TRI5_NC)
        for (int e = lane; e < nT * 4; e += 64) DGT[e] = quad_meta[e];
    // between two transforms the chain wave re-ranks its x and starts the next sweep with rank 0: the indices and the two
    // constants it needs come from LDS tables filled once (global loads there are dependent round trips on the
    // critical path) -- PRM[t][r]: where rank r of transform t goes (rank of transform t - 1, or the feature for t = 0),
    // B3T[t]: (shift, raw log-scale) of rank 0.  x alternates between two arrays; the burst wave zeroes the idle one.
    int* PRM = DGT + DG_WORDS(&m);
    float* B3T = reinterpret_cast<float*>(PRM + T * Dp);
    float* XB = reinterpret_cast<float*>(DGT + ((DG_WORDS(&m) + T * Dp + 2 * T + 3) & ~3));      // (16-byte aligned)
    for (int e = threadIdx.x; e < T * D; e += 64 * (TRI5_NC + 1)) {
        const int tt = e / D, r = e - tt * D;
        const int feat = feat_of_rank[tt * D + r];
        PRM[tt * Dp + r] = tt > 0 ? rank_of_feat[(tt - 1) * D + feat] : feat;
    }
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

    for (int t = T - 1; t >= 0; --t) {
        const float* blk = m.packed + (size_t)t * m.pk_per_transform;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)blk, 0, blk_bytes, 0x00020000);
        float* X = xsel ? XB : XA;                     // zero on entry
        float* Xidle = xsel ? XA : XB;                 // the previous transform's x: re-ranked already, zeroed below
        xsel ^= 1;

        if (wv == TRI5_NC) {
            if (t != T - 1) {
                float4* z4 = reinterpret_cast<float4*>(Xidle);
                for (int e = lane; e < (Dp * 16) >> 2; e += 64) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            // ------------------------------------------------------------------ BURST wave
            // Two fragment sets, used alternately: while tile Tt is being prepared from one set, the fragments
            // of tile Tt+1 are already on their way into the other (nothing else hides their L2 latency here).
            float4 pfA1[PK4], pfA2[PK4], pfB1[PK4], pfB2[PK4];
            float4 lA1, lA2, lB1, lB2, bA1, bA2, bB1, bB2;
            int4 dA, dB;
            bool natural_end = true;
            f32x4 oN[MAXO];                                // output-layer partials of the walker set (right-looking)
#pragma unroll
            for (int O = 0; O < MAXO; ++O) {
                const float4 bb = (O < nOT) ? bload4(rs, vo_q, oB3 + 64 * O) : make_float4(0.f, 0.f, 0.f, 0.f);
                oN[O][0] = bb.x; oN[O][1] = bb.y; oN[O][2] = bb.z; oN[O][3] = bb.w;
            }
#define SW_PREFETCH(TT, P1, P2)                                                                                  \
            switch ((TT) < PK4 ? (TT) : PK4) {                                                                    \
                case 1: prefetch_tile<1>(P1, P2, rs, vo_lane, oF1 + (TT) * nT * 1024, oF2 + (TT) * nT * 1024); break; \
                case 2: prefetch_tile<2>(P1, P2, rs, vo_lane, oF1 + (TT) * nT * 1024, oF2 + (TT) * nT * 1024); break; \
                case 3: prefetch_tile<3>(P1, P2, rs, vo_lane, oF1 + (TT) * nT * 1024, oF2 + (TT) * nT * 1024); break; \
                case 4: prefetch_tile<4>(P1, P2, rs, vo_lane, oF1 + (TT) * nT * 1024, oF2 + (TT) * nT * 1024); break; \
                case 5: prefetch_tile<5>(P1, P2, rs, vo_lane, oF1 + (TT) * nT * 1024, oF2 + (TT) * nT * 1024); break; \
                case 6: prefetch_tile<6>(P1, P2, rs, vo_lane, oF1 + (TT) * nT * 1024, oF2 + (TT) * nT * 1024); break; \
                case 7: prefetch_tile<7>(P1, P2, rs, vo_lane, oF1 + (TT) * nT * 1024, oF2 + (TT) * nT * 1024); break; \
                case 8: prefetch_tile<8>(P1, P2, rs, vo_lane, oF1 + (TT) * nT * 1024, oF2 + (TT) * nT * 1024); break; \
                default: break;                                                                                  \
            }
#define SW_BURST(NK, P1, P2, AA1, AA2, CC1, CC2, HH0, HH1)                                                       \
            switch ((NK) < PK4 ? (NK) : PK4) {                                                                    \
                case 1: burst_tile_open<1>(AA1, AA2, CC1, CC2, P1, P2, HH0, HH1, lane); break;                   \
                case 2: burst_tile_open<2>(AA1, AA2, CC1, CC2, P1, P2, HH0, HH1, lane); break;                   \
                case 3: burst_tile_open<3>(AA1, AA2, CC1, CC2, P1, P2, HH0, HH1, lane); break;                   \
                case 4: burst_tile_open<4>(AA1, AA2, CC1, CC2, P1, P2, HH0, HH1, lane); break;                   \
                case 5: burst_tile_open<5>(AA1, AA2, CC1, CC2, P1, P2, HH0, HH1, lane); break;                   \
                case 6: burst_tile_open<6>(AA1, AA2, CC1, CC2, P1, P2, HH0, HH1, lane); break;                   \
                case 7: burst_tile_open<7>(AA1, AA2, CC1, CC2, P1, P2, HH0, HH1, lane); break;                   \
                case 8: burst_tile_open<8>(AA1, AA2, CC1, CC2, P1, P2, HH0, HH1, lane); break;                   \
                default: break;                                                                                  \
            }
            // everything tile TT needs from global memory: fragments against tiles 0..TT-1 (the last one, K = TT-1,
            // once more in registers of its own: a run-time index into the set would go through scratch), biases,
            // degree words
#define FETCH_TILE(TT, P1, P2, L1, L2, Bb1, Bb2, DGW)                                                            \
            if ((TT) < nT) {                                                                                     \
                SW_PREFETCH(TT, P1, P2)                                                                          \
                if ((TT) > 0) { L1 = bload4(rs, vo_lane, oF1 + ((TT) * nT + (TT) - 1) * 1024);                    \
                                L2 = bload4(rs, vo_lane, oF2 + ((TT) * nT + (TT) - 1) * 1024); }                  \
                Bb1 = bload4(rs, vo_q, oB1 + 64 * (TT)); Bb2 = bload4(rs, vo_q, oB2 + 64 * (TT));                \
                DGW = *reinterpret_cast<const int4*>(DGT + 4 * (TT));                                            \
            }
#define HELPER_TILE(TT, P1, P2, L1, L2, Bb1, Bb2, DGW, NP1, NP2, NL1, NL2, NB1, NB2, NDG)                         \
            {                                                                                                    \
                const int Tt = (TT);                                                                             \
                FETCH_TILE(Tt + 1, NP1, NP2, NL1, NL2, NB1, NB2, NDG)                                            \
                /* same order of additions as tri4: partial sums (a: x,z terms; c: y,w terms) over the first      */ \
                /* min(Tt, PK4) K tiles, folded, then the K tiles beyond PK4 one after the other                  */ \
                f32x4 a1[TRI5_NC], a2[TRI5_NC], c1[TRI5_NC], c2[TRI5_NC];                                        \
                _Pragma("unroll") for (int c = 0; c < TRI5_NC; ++c) {                                            \
                    a1[c][0] = Bb1.x; a1[c][1] = Bb1.y; a1[c][2] = Bb1.z; a1[c][3] = Bb1.w;                      \
                    a2[c][0] = Bb2.x; a2[c][1] = Bb2.y; a2[c][2] = Bb2.z; a2[c][3] = Bb2.w;                      \
                    c1[c] = f32x4{0.f, 0.f, 0.f, 0.f}; c2[c] = f32x4{0.f, 0.f, 0.f, 0.f};                        \
                }                                                                                                \
                /* everything that is final while the chains of tile Tt-1 still run: K <= Tt-2 */                \
                _Pragma("unroll") for (int c = 0; c < TRI5_NC; ++c) {                                            \
                    const float* h0_ = smem + (size_t)c * set_floats + 2 * Dp * 16;                              \
                    const float* h1_ = h0_ + Hp * 16;                                                            \
                    SW_BURST(Tt - 1, P1, P2, a1[c], a2[c], c1[c], c2[c], h0_, h1_)                               \
                    if (Tt - 1 >= PK4) {                                                                         \
                        for (int r = 0; r < 4; ++r) { a1[c][r] += c1[c][r]; a2[c][r] += c2[c][r]; }              \
                        /* (flows wider than PK4 + 1 tiles: four K tiles' fragments in flight -- a load at a time  */ \
                        /*  made the burst wave of a 25-tile flow wait an L2 round trip per K tile)                 */ \
                        float4 w1r[4], w2r[4];                                                                   \
                        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                       \
                            const int Kl = min(PK4 + j_, nT - 1);                                                \
                            w1r[j_] = bload4(rs, vo_lane, oF1 + (Tt * nT + Kl) * 1024);                           \
                            w2r[j_] = bload4(rs, vo_lane, oF2 + (Tt * nT + Kl) * 1024);                           \
                        }                                                                                        \
                        for (int K0 = PK4; K0 < Tt - 1; K0 += 4) {                                               \
                            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                   \
                                const int K = K0 + j_;                                                           \
                                const float4 w1 = w1r[j_], w2 = w2r[j_];                                         \
                                const int Kn = min(K + 4, nT - 1);                                               \
                                w1r[j_] = bload4(rs, vo_lane, oF1 + (Tt * nT + Kn) * 1024);                       \
                                w2r[j_] = bload4(rs, vo_lane, oF2 + (Tt * nT + Kn) * 1024);                       \
                                if (K < Tt - 1) {                                                                \
                                    const float4 b1 = *reinterpret_cast<const float4*>(h0_ + (K << 8) + (lane << 2)); \
                                    const float4 b2 = *reinterpret_cast<const float4*>(h1_ + (K << 8) + (lane << 2)); \
                                    a1[c] = MFMA(w1.x, b1.x, a1[c]); a2[c] = MFMA(w2.x, b2.x, a2[c]);            \
                                    a1[c] = MFMA(w1.y, b1.y, a1[c]); a2[c] = MFMA(w2.y, b2.y, a2[c]);            \
                                    a1[c] = MFMA(w1.z, b1.z, a1[c]); a2[c] = MFMA(w2.z, b2.z, a2[c]);            \
                                    a1[c] = MFMA(w1.w, b1.w, a1[c]); a2[c] = MFMA(w2.w, b2.w, a2[c]);            \
                                }                                                                                \
                            }                                                                                    \
                        }                                                                                        \
                    }                                                                                            \
                }                                                                                                \
                /* the output layer's right-looking updates (what the lone wave does after every group) happen HERE,   */ \
                /* once per tile: oN[O] += F3[O][Tt-1] . h2[Tt-1] for every output tile as soon as tile Tt-1 is final --   */ \
                /* the same additions in the same order (bias, tile after tile, four k chunks); fragments requested now   */ \
                int4 dg = DGW;                                                                                   \
                dg.x &= 0xffff; dg.y &= 0xffff; dg.z &= 0xffff; dg.w &= 0xffff;                                  \
                const bool pad_ = dg.x >= D && dg.y >= D && dg.z >= D && dg.w >= D;                              \
                float4 f3p[MAXO];                                                                                \
                _Pragma("unroll") for (int O = 0; O < MAXO; ++O)                                                 \
                    f3p[O] = (Tt > 0 && O < nOT) ? bload4(rs, vo_lane, oF3 + (O * nT + Tt - 1) * 1024)           \
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);                              \
                if (Tt > 0) lds_bar();                                    /* A(Tt-1): tile Tt-1 is final now */   \
                if (pad_) { natural_end = false; break; }                                                       \
                _Pragma("unroll") for (int c = 0; c < TRI5_NC; ++c) {                                            \
                    float* base_ = smem + (size_t)c * set_floats;                                                \
                    if (Tt > 0) {                                                                                \
                        const int K = Tt - 1;                                                                    \
                        const float* h0_ = base_ + 2 * Dp * 16;                                                  \
                        const float* h1_ = h0_ + Hp * 16;                                                        \
                        const float4 b1 = *reinterpret_cast<const float4*>(h0_ + (K << 8) + (lane << 2));        \
                        const float4 b2 = *reinterpret_cast<const float4*>(h1_ + (K << 8) + (lane << 2));        \
                        if (K < PK4) {                                                                           \
                            a1[c] = MFMA(L1.x, b1.x, a1[c]); a2[c] = MFMA(L2.x, b2.x, a2[c]);                    \
                            c1[c] = MFMA(L1.y, b1.y, c1[c]); c2[c] = MFMA(L2.y, b2.y, c2[c]);                    \
                            a1[c] = MFMA(L1.z, b1.z, a1[c]); a2[c] = MFMA(L2.z, b2.z, a2[c]);                    \
                            c1[c] = MFMA(L1.w, b1.w, c1[c]); c2[c] = MFMA(L2.w, b2.w, c2[c]);                    \
                            for (int r = 0; r < 4; ++r) { a1[c][r] += c1[c][r]; a2[c][r] += c2[c][r]; }          \
                        } else {                                                                                 \
                            a1[c] = MFMA(L1.x, b1.x, a1[c]); a2[c] = MFMA(L2.x, b2.x, a2[c]);                    \
                            a1[c] = MFMA(L1.y, b1.y, a1[c]); a2[c] = MFMA(L2.y, b2.y, a2[c]);                    \
                            a1[c] = MFMA(L1.z, b1.z, a1[c]); a2[c] = MFMA(L2.z, b2.z, a2[c]);                    \
                            a1[c] = MFMA(L1.w, b1.w, a1[c]); a2[c] = MFMA(L2.w, b2.w, a2[c]);                    \
                        }                                                                                        \
                    }                                                                                            \
                    float* sp = base_ + 2 * Dp * 16 + 2 * Hp * 16 + 2 * 256 + (p << 4) + (q << 2);                         \
                    *reinterpret_cast<float4*>(sp + 256) = make_float4(a1[c][0], a1[c][1], a1[c][2], a1[c][3]);  \
                    *reinterpret_cast<float4*>(sp + 512) = make_float4(a2[c][0], a2[c][1], a2[c][2], a2[c][3]);  \
                    if (Tt > 0) {                                                                                \
                        const float4 b = *reinterpret_cast<const float4*>(base_ + 2 * Dp * 16 + 2 * Hp * 16 + (((Tt - 1) & 1) << 8) + (lane << 2)); \
                        _Pragma("unroll") for (int O = 0; O < MAXO; ++O) oN[O] = MFMA(f3p[O].x, b.x, oN[O]);      \
                        _Pragma("unroll") for (int O = 0; O < MAXO; ++O) oN[O] = MFMA(f3p[O].y, b.y, oN[O]);      \
                        _Pragma("unroll") for (int O = 0; O < MAXO; ++O) oN[O] = MFMA(f3p[O].z, b.z, oN[O]);      \
                        _Pragma("unroll") for (int O = 0; O < MAXO; ++O) oN[O] = MFMA(f3p[O].w, b.w, oN[O]);      \
                    }                                                                                            \
                    _Pragma("unroll") for (int O = 0; O < MAXO; ++O)                                             \
                        *reinterpret_cast<float4*>(sp + 768 + O * 256) = make_float4(oN[O][0], oN[O][1], oN[O][2], oN[O][3]); \
                }                                                                                                \
                lds_bar();                                                /* B(Tt) */                            \
            }
            lA1 = lA2 = lB1 = lB2 = bA1 = bA2 = bB1 = bB2 = make_float4(0.f, 0.f, 0.f, 0.f);
            dA = dB = make_int4(0, 0, 0, 0);
            FETCH_TILE(0, pfA1, pfA2, lA1, lA2, bA1, bA2, dA)
            for (int T2 = 0; T2 < nT; T2 += 2) {
                HELPER_TILE(T2, pfA1, pfA2, lA1, lA2, bA1, bA2, dA, pfB1, pfB2, lB1, lB2, bB1, bB2, dB)
                if (T2 + 1 >= nT) break;
                HELPER_TILE(T2 + 1, pfB1, pfB2, lB1, lB2, bB1, bB2, dB, pfA1, pfA2, lA1, lA2, bA1, bA2, dA)
            }
#undef HELPER_TILE
#undef FETCH_TILE
#undef SW_BURST
#undef SW_PREFETCH
            if (natural_end) lds_bar();                               // A(nT-1)
        } else {
            // ------------------------------------------------------------------ CHAIN waves
            ChainRot<MAXO> s;
            {
                const float shift = B3T[2 * t], ls = fast_ls(B3T[2 * t + 1]);
                const float xv = (Y[lidx(0, p)] - shift) * fast_exp_neg(ls);
                ladj -= ls;
                if (q == 0) X[lidx(0, p)] = xv;
            }
            WAVE_LDS_FENCE();
            float4 pf0[PX4], pb0;
#define PREFETCH5(TT)                                                                                          \
            {                                                                                                  \
                const int TT_ = (TT);                                                                          \
                _Pragma("unroll") for (int i_ = 0; i_ < PX4; ++i_)                                             \
                    if (i_ < nXT) pf0[i_] = bload4(rs, vo_lane, oF0 + (TT_ * nXT + i_) * 1024);                 \
                pb0 = bload4(rs, vo_q, oB0 + 64 * TT_);                                                        \
            }
            PREFETCH5(0);
            int4 dg_next = *reinterpret_cast<const int4*>(DGT);
            for (int Tt = 0; Tt < nT; ++Tt) {
                int4 dg = dg_next;
                dg.x &= 0xffff; dg.y &= 0xffff; dg.z &= 0xffff; dg.w &= 0xffff;
                if (dg.x >= D && dg.y >= D && dg.z >= D && dg.w >= D) break;       // padding tiles
                const int pat = 1 | ((dg.y != dg.x) << 1) | ((dg.z != dg.y) << 2) | ((dg.w != dg.z) << 3);
                {
                    const bool ny = dg.y != dg.x, nz = dg.z != dg.y, nw = dg.w != dg.z;
                    s.g[0] = dg.x;
                    s.g[1] = ny ? dg.y : (nz ? dg.z : (nw ? dg.w : D));
                    s.g[2] = ny ? (nz ? dg.z : (nw ? dg.w : D)) : ((nz && nw) ? dg.w : D);
                    s.g[3] = (ny && nz && nw) ? dg.w : D;
                }
                long long* pf = (pa.prof && blockIdx.x == 0) ? pa.prof + ((size_t)(T - 1 - t) * nT + Tt) * 8 : nullptr;
                if (pf && lane == 0) pf[0] = clock64();
                if (!(TRI5_ABL & 16) || Tt == 0) {     // (ablation 16: the chain fragments of the transform's first tile for all)
                const int soD1 = oF1 + (Tt * nT + Tt) * 1024, soD2 = oF2 + (Tt * nT + Tt) * 1024;
                s.wt1 = bload4(rs, vo_T, soD1);
                s.wt2 = bload4(rs, vo_T, soD2);
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const int g_even = s.g[2 * sl], g_odd = s.g[2 * sl + 1];
                    const int gsel = (lane & 2) ? g_odd : g_even;
                    const bool ok = gsel < D;
                    const int gg = ok ? gsel : 0;
                    const int vo = ((((gg >> 3) * nT) << 6) + (q << 4) + 2 * (gg & 7) + (lane & 1)) << 4;
                    const float4 v = bload4(rs, vo, oF3 + Tt * 1024);
                    s.wo[sl] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int gg = s.g[i] < D ? s.g[i] : 0;
#pragma unroll
                    for (int jt = i + 1; jt < 4; ++jt)
                        s.w0r[i][jt] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                            rs, q << 2, oW0 + (gg * Hp + 16 * Tt + 4 * jt) * 4, 0));
                }
                }
                if (!(TRI5_ABL & 0x2000) || Tt == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) s.yv[i] = Y[lidx(s.g[i] < D ? s.g[i] : 0, p)];
                }
                if (pf && lane == 0) pf[1] = clock64();

                // layer 0 against x (final up to the previous tile's ranks)
                f32x4 a0;
                a0[0] = pb0.x; a0[1] = pb0.y; a0[2] = pb0.z; a0[3] = pb0.w;
                if (!(TRI5_ABL & 0x1000) || Tt == 0) {
#pragma unroll
                for (int i = 0; i < PX4; ++i) {
                    if (i < nXT) {
                        const float4 b = *reinterpret_cast<const float4*>(X + (i << 8) + (lane << 2));
                        a0 = MFMA(pf0[i].x, b.x, a0); a0 = MFMA(pf0[i].y, b.y, a0);
                        a0 = MFMA(pf0[i].z, b.z, a0); a0 = MFMA(pf0[i].w, b.w, a0);
                    }
                }
                }
                for (int Xt = PX4; Xt < nXT; ++Xt) {
                    const float4 a = bload4(rs, vo_lane, oF0 + (Tt * nXT + Xt) * 1024);
                    const float4 b = *reinterpret_cast<const float4*>(X + (Xt << 8) + (lane << 2));
                    a0 = MFMA(a.x, b.x, a0); a0 = MFMA(a.y, b.y, a0); a0 = MFMA(a.z, b.z, a0); a0 = MFMA(a.w, b.w, a0);
                }
                {
                    float* sp = S + (p << 4) + (q << 2);
                    *reinterpret_cast<float4*>(sp) = make_float4(a0[0], a0[1], a0[2], a0[3]);
                }
                if (pf && lane == 0) pf[2] = clock64();
                lds_bar();                                            // B(Tt): layers 1/2 of this tile are staged
                if (pf && lane == 0) pf[3] = clock64();
                if (!(TRI5_ABL & 0x800) || Tt == 0) {
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {
                    s.a0[jt] = S[(p << 4) + (jt << 2) + q];
                    s.p1[jt] = S[256 + (p << 4) + (jt << 2) + q];
                    s.p2[jt] = S[512 + (p << 4) + (jt << 2) + q];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int gg = s.g[i] < D ? s.g[i] : 0;
                    s.po[i] = *reinterpret_cast<const float2*>(SO + (gg >> 3) * 256 + (p << 4) + 2 * (gg & 7));
                }
                }
                s.acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
                s.acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
                s.outR[0] = f32x4{0.f, 0.f, 0.f, 0.f};
                s.outR[1] = f32x4{0.f, 0.f, 0.f, 0.f};
                chain_tile_begin(s, X, S, D, q, p, lane);
                if (pf && lane == 0) pf[4] = clock64();
                // the next tile's layer-0 operands and degree words are requested in the shadows of the first group's hops
                const bool more = Tt + 1 < nT;
                auto ahead = [&](auto gi_, auto hop_) {
                    constexpr int G = decltype(gi_)::value, HP = decltype(hop_)::value;
                    if constexpr (G == 0) {
                        if (more && !(TRI5_ABL & 0x100)) {
                            if constexpr (HP < PX4) { if (HP < nXT) pf0[HP] = bload4(rs, vo_lane, oF0 + ((Tt + 1) * nXT + HP) * 1024); }
                            if constexpr (HP == 2) {
                                pb0 = bload4(rs, vo_q, oB0 + 64 * (Tt + 1));
                                dg_next = *reinterpret_cast<const int4*>(DGT + 4 * (Tt + 1));
                            }
                        }
                    }
                };
                static_assert(PX4 <= 3, "one layer-0 fragment per hop of the first group");
                if (pat == 15 || (TRI5_ABL & 0x200)) {             // four single-quad groups: the common tile
                    chain_group_rot<15, 0, 4, MAXO, TRI5_ABL | 1>(s, H0, H1, X, Tt, D, nOT, q, p, ladj, H2, ahead);
                } else
                switch (pat) {
#define CASE(P) case P: chain_group_rot<P, 0, 4, MAXO, TRI5_ABL | 1>(s, H0, H1, X, Tt, D, nOT, q, p, ladj, H2, ahead); break;
                    CASE(1) CASE(3) CASE(5) CASE(7) CASE(9) CASE(11) CASE(13)
#undef CASE
                    default: break;
                }
                chain_flush(s, ladj);
                if (pf && lane == 0) pf[5] = clock64();
                lds_bar();                                            // A(Tt): this tile is final
                if (pf && lane == 0) pf[6] = clock64();
            }
#undef PREFETCH5
        }
        const bool last = (t == 0);
        if (wv < TRI5_NC) {
            const int* prm = PRM + t * Dp;
            for (int e = lane; e < D * 16; e += 64) {
                const int r = e >> 4, pp = e & 15;
                const float v = X[lidx(r, pp)];
                const int tgt = prm[r];
                if (!last) Y[lidx(tgt, pp)] = v;
                else if (row0 + pp < n) out[(row0 + pp) * D + tgt] = v;
            }
            if (!last)
                for (int e = lane; e < (Dp - D) * 16; e += 64) Y[lidx(D + (e >> 4), e & 15)] = 0.0f;
        }
        __syncthreads();
        if (last) xsel ^= 1;                           // (xsel names the array of the LAST transform again: the epilogue reads it)
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
    static const int mode = getenv("PMC_INVERSE_DUO") ? atoi(getenv("PMC_INVERSE_DUO")) : -1;
    return mode;
}

static bool tri5_wanted(const pmc_maf_t* m, int64_t n) {
    const int mode = tri5_mode();
    if (mode >= 0) return mode != 0;
    const int maxo = m->nOT <= 4 ? 4 : 8;
    const size_t lds1 = (size_t)(2 * m->Dp * 16 + 2 * m->Hp * 16 + 3 * 256 + maxo * 256 + DG_WORDS(m)) * sizeof(float);   // one walker set
    const size_t lds5 = (size_t)(2 * m->Dp * 16 + 2 * m->Hp * 16 + 5 * 256 + maxo * 256 + TRI5_TABLE_WORDS(m)) * sizeof(float);   // (two-wave sweep)
    if (lds5 * TRI5_NC > 160 * 1024) return false;
    // rounds a launch needs: workgroups resident per CU are bounded by the LDS (both kernels keep one set's tiles
    // per chain wave) and by the SIMDs (one 256-register wave each: 4 lone waves or 4 / (TRI5_NC + 1) groups).
    // The two-wave kernel takes ~0.75 of the lone wave's time per round.
    // (512 bytes of slack per workgroup: three workgroups of 54 560 bytes do not share a CU's 163 840 in practice)
    const int64_t by_lds4 = (int64_t)((160 * 1024) / (lds1 + 512)), by_lds5 = (int64_t)((160 * 1024) / (lds5 * TRI5_NC + 512));
    const int64_t res4 = 256 * (by_lds4 < 4 ? by_lds4 : 4);
    const int64_t wg5 = by_lds5 < 4 / (TRI5_NC + 1) ? by_lds5 : 4 / (TRI5_NC + 1);
    const int64_t res5 = 256 * TRI5_NC * wg5;
    if (res4 < 1 || res5 < 1) return false;
    const int64_t sets = (n + 15) / 16;
    const int64_t r4 = (sets + res4 - 1) / res4, r5 = (sets + res5 - 1) / res5;
    return 3 * r5 < 4 * r4;
}

// which of the two D <= 64 sweeps PMC_INVERSE_AUTO launches for n rows (bench.py names the kernel it times with it)
extern "C" int pmc_debug_inverse_uses_duo(const pmc_maf_t* m, int64_t n) {
    if (!m || m->n_out != 2 || !m->tri_ok || m->nOT > 8 || m->D > 64) return 0;
    return tri5_wanted(m, n) ? 1 : 0;
}

// same contract as pmc_launch_propose_inverse_tri4 / pmc_launch_inverse_tri4 (pa == nullptr: plain inverse of z)
static int launch_tri5(const ProposeArgs* pa, const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n,
                       hipStream_t stream) {
    if (m->n_out != 2 || !m->tri_ok || m->nOT > 8 || m->D > 64) return -1;
    if (m->pk_per_transform * 4 > 0x7fffffffLL) return -1;
    const int maxo = m->nOT <= 4 ? 4 : 8;
    const size_t lds = ((size_t)TRI5_NC * (2 * m->Dp * 16 + 2 * m->Hp * 16 + 5 * 256 + maxo * 256) + TRI5_TABLE_WORDS(m)) * sizeof(float);
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

// measurement only (scripts/profile_tri5.py): cycle stamps of the chain wave of workgroup 0 -- prof[transform * nT + tile][8]
extern "C" int pmc_debug_tri5_profile(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n, long long* prof,
                                      void* stream) {
    ProposeArgs pa{};
    pa.prof = prof;
    return launch_tri5(&pa, m, z, x, ladj, n, (hipStream_t)stream) < 0 ? pmc_fail("pmc_debug_tri5_profile: flow not covered") : 0;
}

// Triangular-sweep inverse of the neural spline flows as TWO wavefronts per 16 walkers (pocomc/mcmc.py:88 ->
// flow.py:116-132 with flow = nsf3 | nsf6 | nsf12): the spline counterpart of maf_inverse_tri5_kernel.
//
// The lone-wave spline sweep (maf_inverse_tri_nsf.hip) spends its time where the affine one did before it was split --
// 272 us for 7008 walkers of nsf3 @ D = 32, of which (timing-only builds, scripts/abl_nsf.sh) ~100 us are the rank's
// left-looking output product (23 parameters = two 16-row tiles against every final h2 tile: 8 (Tt + 1) MFMAs per
// rank), ~65 us the hidden chain with its left-looking bursts, ~53 us the spline solves, ~55 us everything else.  As
// in the affine sweep the work is cut by WHEN its inputs exist:
//   * CHAIN wave: per degree group the three hidden hops on transposed accumulators (maf_chain_rot.h: one MFMA per
//     layer and quad), its share of the NEXT tile's hidden pre-activations (right-looking: accN1 / accN2 / a0N), the
//     part of the rank's 23 parameters that comes from the previous and the own hidden tile (8 + 2 (quads so far)
//     MFMAs, fragments requested one group ahead), the exchange through a 2 KB LDS panel and the spline solve;
//   * BURST wave, a whole tile ahead: the hidden layers' left-looking products against tiles <= Tt-2 (as in tri5: f0c
//     against x, f1 / f2 against h0 / h1, into transposed staging) AND, for each of the tile's ranks, bias + the two
//     output tiles against h2 tiles <= Tt-2 (into a staged partial the chain's accumulators start from: both waves hold
//     a 16 x 16 product in the same lane layout, so staging is one 16-byte write and read per lane at the same address).
// One LDS-only barrier per tile: E(Tt) = "tile Tt is final, the staging of tile Tt+1 is complete".
#include <stdlib.h>
#include "maf_chain_rot.h"
#include "rqs.h"

#ifndef NSF2_ABL
#define NSF2_ABL 0                 // timing experiments only (scripts/abl_nsf.sh): results are wrong when != 0
#endif                             // 1 no spline solve, 2 no output MFMAs on the chain, 4 burst: no output partials, 8 nor their loads, 16 chain: no output fragment requests, 32 burst: no hidden products
#define NSF2_PK 8                  // K tiles of the hidden bursts held in registers
#define NSF2_PX 4                  // x tiles of the layer-0 product held in registers (D <= 64)
#define NSF2_PO 8                  // K tiles of a rank's output partial held in registers
#define NSF2_OOB 0x40000000        // a lane offset beyond every image: the bounds-checked load returns zeros
#define NSF2_STAGE_FLOATS (3 * 256)                 // hidden staging S0 | S1 | S2 (transposed, [lane][4])
#define NSF2_PART_FLOATS (4 * 2 * 256)              // output staging [group][half][lane][4]
#define NSF2_TT_WORDS(m) (((m)->nT + 2) * 8)        // per-tile table: ranks (word 0 also the pattern), x / y byte offsets
#define NSF2_LDS_FLOATS(m) (3 * (m)->Dp * 16 + 3 * (m)->Hp * 16 + 2 * NSF2_STAGE_FLOATS + 2 * NSF2_PART_FLOATS + 16 * 32 + 16 * 24 + \
                            ((NSF2_TT_WORDS(m) + (m)->T * (m)->Dp + 3) & ~3))

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 nbload4(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ f32x4 as_acc(const float4& v) { f32x4 r; r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w; return r; }

// the chain's hidden-layer operands of a tile (static data, requested a tile ahead) and its table words
struct NsfHid {
    float4 wt1, wt2;               // diagonal tile of layers 1 / 2, rows transposed (chain_vo_T); component c = K chunk c
    float4 wn1, wn2;               // block (Tt + 1, Tt), transposed
    float4 w0o[4];                 // [group].jt: W0[row q of quad jt of this tile][the rank the group produces]
    float4 w0N[4];                 // the same against the next tile's quads
    int g[4], xy[4], pat;
};
// a rank's two output tiles against the previous (fp) and the own (fc) hidden tile, natural fragments
struct NsfOut { float4 fp0, fp1, fc0, fc1; };

struct NsfChain {
    float a0[4], p1[4], p2[4];
    f32x4 acc1, acc2, accN1, accN2;
    float a0N[4];
    float h0s[4], h1s[4], h2s[4];  // this tile's activations, row q of every quad
    float h2p[4];                  // the previous tile's h2 (B operands of the fp products)
};

template <int N, class F>
__device__ __forceinline__ void nsf_for(F&& f) {
    if constexpr (N > 0) { nsf_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

// Groups I .. of a tile with quad pattern PAT, one after the other, straight-line.
//   ob[I & 1]: this group's output fragments; `ahead(I)` requests the next group's into ob[(I + 1) & 1] (and whatever
//   else the caller wants in flight).
template <int PAT, int I, class AH>
__device__ __forceinline__ void nsf_group(NsfChain& s, const NsfHid& f, NsfOut (&ob)[2], const float* part, float* X, const float* Y,
                                          float* PAR, float* TAB, int D, int q, int p, int lane, float& ladj, const AH& ahead) {
    constexpr int NG = pat_ngroups(PAT);
    if constexpr (I < NG) {
        constexpr int c0 = pat_start(PAT, I), c1 = pat_end(PAT, I);
        {
            const NsfOut& o = ob[I & 1];
            ahead(std::integral_constant<int, I>{});
            CHAIN_FENCE();
            // ---- the rank's parameters, part that does not wait for this group's hops: staged partial (bias + h2 tiles
            // <= Tt-2, burst wave) + previous tile + the own tile's earlier quads
            f32x4 o0 = as_acc(*reinterpret_cast<const float4*>(part + (2 * I) * 256 + (lane << 2)));
            f32x4 o1 = as_acc(*reinterpret_cast<const float4*>(part + (2 * I + 1) * 256 + (lane << 2)));
            if (!(NSF2_ABL & 2)) {
                o0 = MFMA(o.fp0.x, s.h2p[0], o0); o1 = MFMA(o.fp1.x, s.h2p[0], o1);
                o0 = MFMA(o.fp0.y, s.h2p[1], o0); o1 = MFMA(o.fp1.y, s.h2p[1], o1);
                o0 = MFMA(o.fp0.z, s.h2p[2], o0); o1 = MFMA(o.fp1.z, s.h2p[2], o1);
                o0 = MFMA(o.fp0.w, s.h2p[3], o0); o1 = MFMA(o.fp1.w, s.h2p[3], o1);
#pragma unroll
                for (int c = 0; c < c0; ++c) { o0 = MFMA(comp(o.fc0, c), s.h2s[c], o0); o1 = MFMA(comp(o.fc1, c), s.h2s[c], o1); }
            }
            CHAIN_FENCE();
            float h0[4], h1[4], h2[4];
            // ---------------------------------------------------------------- hop 1
#pragma unroll
            for (int c = c0; c <= c1; ++c) { h0[c] = fmaxf(s.a0[c], 0.0f); s.h0s[c] = h0[c]; }
            CHAIN_FENCE();
#pragma unroll
            for (int c = c0; c <= c1; ++c) s.acc1 = MFMA(comp(f.wt1, c), h0[c], s.acc1);
            CHAIN_FENCE();
#pragma unroll
            for (int c = c0; c <= c1; ++c) s.accN1 = MFMA(comp(f.wn1, c), h0[c], s.accN1);
            CHAIN_FENCE();
#pragma unroll
            for (int c = c0; c <= c1; ++c) { h1[c] = fmaxf((s.acc1[c] + s.p1[c]) + h0[c], 0.0f); s.h1s[c] = h1[c]; }
            CHAIN_FENCE();
            // ---------------------------------------------------------------- hop 2
#pragma unroll
            for (int c = c0; c <= c1; ++c) s.acc2 = MFMA(comp(f.wt2, c), h1[c], s.acc2);
            CHAIN_FENCE();
#pragma unroll
            for (int c = c0; c <= c1; ++c) s.accN2 = MFMA(comp(f.wn2, c), h1[c], s.accN2);
            CHAIN_FENCE();
#pragma unroll
            for (int c = c0; c <= c1; ++c) { h2[c] = fmaxf((s.acc2[c] + s.p2[c]) + h1[c], 0.0f); s.h2s[c] = h2[c]; }
            CHAIN_FENCE();
            // ---------------------------------------------------------------- hop 3: the group's own quads
            if (!(NSF2_ABL & 2)) {
#pragma unroll
                for (int c = c0; c <= c1; ++c) { o0 = MFMA(comp(o.fc0, c), h2[c], o0); o1 = MFMA(comp(o.fc1, c), h2[c], o1); }
            }
            CHAIN_FENCE();
            // ---------------------------------------------------------------- every lane gets its row's 23 values; spline
            const float yv = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(Y) + (p << 4) + f.xy[I]);
            float xg, l;
            if (NSF2_ABL & 1) {
                float* pr = PAR + (p << 5) + (q << 2);
                *reinterpret_cast<float4*>(pr) = make_float4(o0[0], o0[1], o0[2], o0[3]);
                *reinterpret_cast<float4*>(pr + 16) = make_float4(o1[0], o1[1], o1[2], o1[3]);
                WAVE_LDS_FENCE();
                xg = yv + PAR[(p << 5)] + PAR[(p << 5) + 22]; l = PAR[(p << 5) + 8];
            } else {
                rqs_inverse_split(o0, o1, PAR + (p << 5), q, yv, xg, l);
            }
            if (q == 0) *reinterpret_cast<float*>(reinterpret_cast<char*>(X) + (p << 4) + f.xy[I]) = xg;
            ladj -= l;
            WAVE_LDS_FENCE();
            // ---------------------------------------------------------------- rank-1 updates of layer 0: own tile, next tile
#pragma unroll
            for (int jt = c1 + 1; jt < 4; ++jt) s.a0[jt] = fmaf(comp(f.w0o[I], jt), xg, s.a0[jt]);
            s.a0N[0] = fmaf(f.w0N[I].x, xg, s.a0N[0]); s.a0N[1] = fmaf(f.w0N[I].y, xg, s.a0N[1]);
            s.a0N[2] = fmaf(f.w0N[I].z, xg, s.a0N[2]); s.a0N[3] = fmaf(f.w0N[I].w, xg, s.a0N[3]);
            CHAIN_FENCE();
            nsf_group<PAT, I + 1, AH>(s, f, ob, part, X, Y, PAR, TAB, D, q, p, lane, ladj, ahead);
        }
    }
}

// BURST wave: bias + the two output tiles against the final h2 tiles 0 .. NK-1 for the (up to four) ranks of a tile, into
// the staged partials the chain's accumulators start from.  NK is a compile-time constant (dispatched once per tile): a
// rank's 2 NK fragment loads are issued while the previous rank's MFMAs run, and nothing is requested that is not used
// (an out-of-range request costs the vector memory pipe as much as a real one).
template <int NK>
__device__ __forceinline__ void nsf_out_partials(__amdgpu_buffer_rsrc_t rs, const int tb_f3i, const int tb_b3i, const int nT, const int D,
                                                 const int g0, const int g1, const int g2, const int g3, const float* H2,
                                                 float* dst, const int lane, const int vo_lane, const int vo_q) {
    constexpr int NF = NK > 0 ? NK : 1;
    float4 fa0[NF], fa1[NF], fb0[NF], fb1[NF], ba0, ba1, bb0, bb1;
    auto fetch = [&](const int g, float4 (&F0)[NF], float4 (&F1)[NF], float4& B0, float4& B1) {
        const bool lv = g < D;
        const int gg = lv ? g : 0;
        const int so = tb_f3i + gg * 2 * nT * 1024;
        const int vo = lv ? vo_lane : NSF2_OOB;
        if (!(NSF2_ABL & 8))
#pragma unroll
        for (int i = 0; i < NK; ++i) { F0[i] = nbload4(rs, vo, so + i * 1024); F1[i] = nbload4(rs, vo, so + (nT + i) * 1024); }
        B0 = nbload4(rs, lv ? vo_q : NSF2_OOB, tb_b3i + gg * 128);
        B1 = nbload4(rs, lv ? vo_q : NSF2_OOB, tb_b3i + gg * 128 + 64);
    };
    auto comp_store = [&](const float4 (&F0)[NF], const float4 (&F1)[NF], const float4& B0, const float4& B1, float* d) {
        f32x4 o0 = as_acc(B0), o1 = as_acc(B1);
        if (!(NSF2_ABL & 4)) {
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const float4 b = *reinterpret_cast<const float4*>(H2 + (i << 8) + (lane << 2));
                o0 = MFMA(F0[i].x, b.x, o0); o1 = MFMA(F1[i].x, b.x, o1);
                o0 = MFMA(F0[i].y, b.y, o0); o1 = MFMA(F1[i].y, b.y, o1);
                o0 = MFMA(F0[i].z, b.z, o0); o1 = MFMA(F1[i].z, b.z, o1);
                o0 = MFMA(F0[i].w, b.w, o0); o1 = MFMA(F1[i].w, b.w, o1);
            }
        }
        *reinterpret_cast<float4*>(d + (lane << 2)) = make_float4(o0[0], o0[1], o0[2], o0[3]);
        *reinterpret_cast<float4*>(d + 256 + (lane << 2)) = make_float4(o1[0], o1[1], o1[2], o1[3]);
    };
    fetch(g0, fa0, fa1, ba0, ba1);
    fetch(g1, fb0, fb1, bb0, bb1);
    comp_store(fa0, fa1, ba0, ba1, dst);
    fetch(g2, fa0, fa1, ba0, ba1);
    comp_store(fb0, fb1, bb0, bb1, dst + 512);
    fetch(g3, fb0, fb1, bb0, bb1);
    comp_store(fa0, fa1, ba0, ba1, dst + 1024);
    comp_store(fb0, fb1, bb0, bb1, dst + 1536);
}

// the same for flows with more than NSF2_PO + 1 hidden tiles: fragments streamed K tile by K tile
__device__ __forceinline__ void nsf_out_partials_wide(__amdgpu_buffer_rsrc_t rs, const int tb_f3i, const int tb_b3i, const int nT, const int D,
                                                      const int nK, const int g, const float* H2, float* d, const int lane,
                                                      const int vo_lane, const int vo_q) {
    const bool lv = g < D;
    const int gg = lv ? g : 0;
    f32x4 o0 = as_acc(nbload4(rs, lv ? vo_q : NSF2_OOB, tb_b3i + gg * 128)), o1 = as_acc(nbload4(rs, lv ? vo_q : NSF2_OOB, tb_b3i + gg * 128 + 64));
    if (lv) {
        for (int K = 0; K < nK; ++K) {
            const float4 w0 = nbload4(rs, vo_lane, tb_f3i + (gg * 2 * nT + K) * 1024);
            const float4 w1 = nbload4(rs, vo_lane, tb_f3i + ((gg * 2 + 1) * nT + K) * 1024);
            const float4 b = *reinterpret_cast<const float4*>(H2 + (K << 8) + (lane << 2));
            o0 = MFMA(w0.x, b.x, o0); o1 = MFMA(w1.x, b.x, o1);
            o0 = MFMA(w0.y, b.y, o0); o1 = MFMA(w1.y, b.y, o1);
            o0 = MFMA(w0.z, b.z, o0); o1 = MFMA(w1.z, b.z, o1);
            o0 = MFMA(w0.w, b.w, o0); o1 = MFMA(w1.w, b.w, o1);
        }
    }
    *reinterpret_cast<float4*>(d + (lane << 2)) = make_float4(o0[0], o0[1], o0[2], o0[3]);
    *reinterpret_cast<float4*>(d + 256 + (lane << 2)) = make_float4(o1[0], o1[1], o1[2], o1[3]);
}

template <int FM>
__global__ __launch_bounds__(128) void maf_inverse_nsf2_kernel(pmc_maf_t m, const float* __restrict__ in, float* __restrict__ out,
                                                               float* __restrict__ ladj_out, int64_t n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, p = lane & 15;
    const int D = m.D, Dp = m.Dp, Hp = m.Hp, T = m.T, nT = m.nT, nXT = m.nXT, nOT = m.nOT;
    const int nTl = __builtin_amdgcn_readfirstlane(m.meta[7]);
    const int64_t row0 = (int64_t)blockIdx.x * 16;
    float* Y = smem;
    float* XA = Y + Dp * 16;
    float* XB = XA + Dp * 16;
    float* H0 = XB + Dp * 16;
    float* H1 = H0 + Hp * 16;
    float* H2 = H1 + Hp * 16;
    float* STG = H2 + Hp * 16;                 // two hidden staging buffers (tile parity)
    float* PART = STG + 2 * NSF2_STAGE_FLOATS; // two output staging buffers (tile parity)
    float* PAR = PART + 2 * NSF2_PART_FLOATS;  // [16 rows][32]: the 23 spline parameters of the current rank
    float* TAB = PAR + 16 * 32;                // [16 rows][24]: knot tables (rqs_inverse_coop)
    int* DGT = reinterpret_cast<int*>(TAB + 16 * 24);
    int* PRM = DGT + NSF2_TT_WORDS(&m);
    const int* feat_of_rank = m.meta + 8;
    const int* rank_of_feat = m.meta + 8 + T * D;
    const int* quad_meta = m.meta + 8 + 2 * T * D;

    // byte offsets of the packed arrays inside one transform's block (maf_spec.py: pk_offsets of the spline image)
    const int oF1 = nT * nXT * 1024;
    const int oF2 = oF1 + nT * nT * 1024;
    const int oF3 = oF2 + nT * nT * 1024;
    const int oW0 = oF3 + nOT * nT * 1024;
    const int oB0 = oW0 + Dp * Hp * 4;
    const int oB3 = oB0 + 3 * Hp * 4;
    const int oF3I = oB3 + nOT * 64;
    const int oB3I = oF3I + D * 2 * nT * 1024;
    const int oCW0 = oB3I + D * 128;
    const int oF0C = oCW0 + nT * 1024;
    const int oB0T = oF0C + nT * nXT * 1024;
    const int oB1T = oB0T + Hp * 4;
    const int oB2T = oB1T + Hp * 4;
    const int blk_bytes = (int)(m.pk_per_transform * 4);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)m.packed, 0, blk_bytes * T, 0x00020000);
    const int vo_lane = lane << 4;
    const int vo_T = chain_vo_T(lane);
    const int vo_q = q << 4;
    auto lds_bar = []() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // per hidden tile 8 words (two rows of "no groups" behind the last tile): the ranks its groups produce (word 0 also
    // the quad pattern << 16), the byte offsets of those ranks' x / y word (walker 0)
    for (int e = threadIdx.x; e < (nT + 2) * 8; e += 128) {
        const int tile = e >> 3, k = e & 7, i = k & 3;
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
            // padding groups (degree >= D: zero weights, zero bias) trail; the last live group absorbs their quads -- the
            // chain then runs live groups only and needs no per-group condition (a condition around a group makes every
            // register it updates a conditional assignment: ~200 copies per group)
            const int ng = 1 + ny + nz + nw;
            const int gs[4] = {dg.x, g1, g2, g3};
            int keep = 0, seen = 0;
            for (int j = 0; j < 4; ++j) {
                if ((pat >> j) & 1) { if (seen < ng && gs[seen] < D) keep |= 1 << j; ++seen; }
            }
            pat = keep ? keep : 1;
        }
        const int gg = g < D ? g : 0;
        DGT[e] = k < 4 ? (g | (i == 0 ? pat << 16 : 0)) : 4 * (((gg >> 4) << 8) + ((gg & 3) << 6) + ((gg >> 2) & 3));
    }
    for (int e = threadIdx.x; e < T * D; e += 128) {
        const int tt = e / D, r = e - tt * D;
        const int feat = feat_of_rank[tt * D + r];
        PRM[tt * Dp + r] = tt > 0 ? rank_of_feat[(tt - 1) * D + feat] : feat;
    }
    {   // x arrays, activations (their padding slots are read against zero weights) and staging start zeroed
        float4* z4 = reinterpret_cast<float4*>(XA);
        const int n4 = (2 * Dp * 16 + 3 * Hp * 16 + 2 * NSF2_STAGE_FLOATS + 2 * NSF2_PART_FLOATS) >> 2;
        for (int e = threadIdx.x; e < n4; e += 128) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (wv == 0) load_rows(Y, in, row0, n, D, Dp, feat_of_rank + (T - 1) * D, lane);
    float ladj = 0.0f;
    int xsel = 0;
    __syncthreads();

    auto take_table = [&](NsfHid& F, const int U) {
        const int4 tg = *reinterpret_cast<const int4*>(DGT + 8 * U);
        const int4 txy = *reinterpret_cast<const int4*>(DGT + 8 * U + 4);
        F.g[0] = __builtin_amdgcn_readfirstlane(tg.x & 0xffff); F.g[1] = __builtin_amdgcn_readfirstlane(tg.y);
        F.g[2] = __builtin_amdgcn_readfirstlane(tg.z); F.g[3] = __builtin_amdgcn_readfirstlane(tg.w);
        F.pat = __builtin_amdgcn_readfirstlane(tg.x >> 16);
        F.xy[0] = __builtin_amdgcn_readfirstlane(txy.x); F.xy[1] = __builtin_amdgcn_readfirstlane(txy.y);
        F.xy[2] = __builtin_amdgcn_readfirstlane(txy.z); F.xy[3] = __builtin_amdgcn_readfirstlane(txy.w);
    };

    if (wv == 1) {
        // ==================================================================================== BURST wave
        // hidden operands of tile TT (two sets, used alternately: tile TT+1's are on their way while TT is prepared)
#define NB_FETCH(TB, TT, P1, P2, XF, Bz0, Bz1, Bz2)                                                               \
        {                                                                                                         \
            const int TT_ = (TT) < nT ? (TT) : nT - 1;                                                            \
            const int so1_ = (TB) + oF1 + TT_ * nT * 1024, so2_ = (TB) + oF2 + TT_ * nT * 1024;                   \
            _Pragma("unroll") for (int i_ = 0; i_ < NSF2_PK; ++i_) {                                              \
                const int vo_ = i_ < TT_ - 1 ? vo_T : NSF2_OOB;                                                   \
                P1[i_] = nbload4(rs, vo_, so1_ + i_ * 1024);                                                      \
                P2[i_] = nbload4(rs, vo_, so2_ + i_ * 1024);                                                      \
            }                                                                                                     \
            _Pragma("unroll") for (int i_ = 0; i_ < NSF2_PX; ++i_)                                                \
                XF[i_] = nbload4(rs, i_ < nXT ? vo_T : NSF2_OOB, (TB) + oF0C + (TT_ * nXT + i_) * 1024);           \
            Bz0 = nbload4(rs, vo_q, (TB) + oB0T + 64 * TT_);                                                      \
            Bz1 = nbload4(rs, vo_q, (TB) + oB1T + 64 * TT_);                                                      \
            Bz2 = nbload4(rs, vo_q, (TB) + oB2T + 64 * TT_);                                                      \
        }
        float4 pA1[NSF2_PK], pA2[NSF2_PK], pB1[NSF2_PK], pB2[NSF2_PK], xA[NSF2_PX], xB[NSF2_PX];
        float4 bA0, bA1, bA2, bB0, bB1, bB2;
        NB_FETCH((T - 1) * blk_bytes, 0, pA1, pA2, xA, bA0, bA1, bA2)
        for (int t = T - 1; t >= 0; --t) {
            const int tb = t * blk_bytes;
            float* X = xsel ? XB : XA;                     // zero on entry
            float* Xidle = xsel ? XA : XB;
            xsel ^= 1;
            if (t != T - 1) {
                float4* z4 = reinterpret_cast<float4*>(Xidle);
                for (int e = lane; e < (Dp * 16) >> 2; e += 64) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#define NB_K(NK, P1, P2, AA1, AA2)                                                                                \
            _Pragma("unroll") for (int i_ = 0; i_ < NSF2_PK; ++i_) {                                              \
                if (i_ < (NK)) {                                                                                  \
                    const float4 b1 = *reinterpret_cast<const float4*>(H0 + (i_ << 8) + (lane << 2));             \
                    const float4 b2 = *reinterpret_cast<const float4*>(H1 + (i_ << 8) + (lane << 2));             \
                    AA1 = MFMA(P1[i_].x, b1.x, AA1); AA2 = MFMA(P2[i_].x, b2.x, AA2);                             \
                    AA1 = MFMA(P1[i_].y, b1.y, AA1); AA2 = MFMA(P2[i_].y, b2.y, AA2);                             \
                    AA1 = MFMA(P1[i_].z, b1.z, AA1); AA2 = MFMA(P2[i_].z, b2.z, AA2);                             \
                    AA1 = MFMA(P1[i_].w, b1.w, AA1); AA2 = MFMA(P2[i_].w, b2.w, AA2);                             \
                }                                                                                                 \
            }
            // prepare tile T1 from set (P1 ...) while the chain runs tile T1-1; request tile T1+1 into set (N1 ...)
#define NB_TILE(TT, P1, P2, XF, Bz0, Bz1, Bz2, N1, N2, NXF, Nz0, Nz1, Nz2)                                         \
            {                                                                                                     \
                const int T1 = (TT);                                                                              \
                const int nK = T1 - 1;                          /* hidden tiles 0 .. T1-2 are final */            \
                const int4 tg_ = *reinterpret_cast<const int4*>(DGT + 8 * T1);                                    \
                const int g0_ = __builtin_amdgcn_readfirstlane(tg_.x & 0xffff), g1_ = __builtin_amdgcn_readfirstlane(tg_.y); \
                const int g2_ = __builtin_amdgcn_readfirstlane(tg_.z), g3_ = __builtin_amdgcn_readfirstlane(tg_.w); \
                NB_FETCH(tb, T1 + 1, N1, N2, NXF, Nz0, Nz1, Nz2)                                                  \
                f32x4 a0 = as_acc(Bz0), a1 = as_acc(Bz1), a2 = as_acc(Bz2);                                       \
                if (!(NSF2_ABL & 32)) NB_K(nK, P1, P2, a1, a2)                                                                          \
                for (int K = NSF2_PK; K < nK; ++K) {            /* flows wider than NSF2_PK + 2 tiles */          \
                    const float4 w1 = nbload4(rs, vo_T, tb + oF1 + (T1 * nT + K) * 1024);                         \
                    const float4 w2 = nbload4(rs, vo_T, tb + oF2 + (T1 * nT + K) * 1024);                         \
                    const float4 b1 = *reinterpret_cast<const float4*>(H0 + (K << 8) + (lane << 2));              \
                    const float4 b2 = *reinterpret_cast<const float4*>(H1 + (K << 8) + (lane << 2));              \
                    a1 = MFMA(w1.x, b1.x, a1); a2 = MFMA(w2.x, b2.x, a2);                                         \
                    a1 = MFMA(w1.y, b1.y, a1); a2 = MFMA(w2.y, b2.y, a2);                                         \
                    a1 = MFMA(w1.z, b1.z, a1); a2 = MFMA(w2.z, b2.z, a2);                                         \
                    a1 = MFMA(w1.w, b1.w, a1); a2 = MFMA(w2.w, b2.w, a2);                                         \
                }                                                                                                 \
                _Pragma("unroll") for (int i_ = 0; i_ < NSF2_PX; ++i_) {                                          \
                    if (i_ < nXT) {                                                                               \
                        const float4 b = *reinterpret_cast<const float4*>(X + (i_ << 8) + (lane << 2));            \
                        a0 = MFMA(XF[i_].x, b.x, a0); a0 = MFMA(XF[i_].y, b.y, a0);                               \
                        a0 = MFMA(XF[i_].z, b.z, a0); a0 = MFMA(XF[i_].w, b.w, a0);                               \
                    }                                                                                             \
                }                                                                                                 \
                float* st_ = STG + (T1 & 1) * NSF2_STAGE_FLOATS;                                                  \
                *reinterpret_cast<float4*>(st_ + (lane << 2)) = make_float4(a0[0], a0[1], a0[2], a0[3]);          \
                *reinterpret_cast<float4*>(st_ + 256 + (lane << 2)) = make_float4(a1[0], a1[1], a1[2], a1[3]);    \
                *reinterpret_cast<float4*>(st_ + 512 + (lane << 2)) = make_float4(a2[0], a2[1], a2[2], a2[3]);    \
                float* pt_ = PART + (T1 & 1) * NSF2_PART_FLOATS;                                                  \
                const int f3_ = tb + oF3I, b3_ = tb + oB3I;                                                       \
                switch (nK < 0 ? 0 : nK) {                                                                        \
                    case 0: nsf_out_partials<0>(rs, f3_, b3_, nT, D, g0_, g1_, g2_, g3_, H2, pt_, lane, vo_lane, vo_q); break; \
                    case 1: nsf_out_partials<1>(rs, f3_, b3_, nT, D, g0_, g1_, g2_, g3_, H2, pt_, lane, vo_lane, vo_q); break; \
                    case 2: nsf_out_partials<2>(rs, f3_, b3_, nT, D, g0_, g1_, g2_, g3_, H2, pt_, lane, vo_lane, vo_q); break; \
                    case 3: nsf_out_partials<3>(rs, f3_, b3_, nT, D, g0_, g1_, g2_, g3_, H2, pt_, lane, vo_lane, vo_q); break; \
                    case 4: nsf_out_partials<4>(rs, f3_, b3_, nT, D, g0_, g1_, g2_, g3_, H2, pt_, lane, vo_lane, vo_q); break; \
                    case 5: nsf_out_partials<5>(rs, f3_, b3_, nT, D, g0_, g1_, g2_, g3_, H2, pt_, lane, vo_lane, vo_q); break; \
                    case 6: nsf_out_partials<6>(rs, f3_, b3_, nT, D, g0_, g1_, g2_, g3_, H2, pt_, lane, vo_lane, vo_q); break; \
                    case 7: nsf_out_partials<7>(rs, f3_, b3_, nT, D, g0_, g1_, g2_, g3_, H2, pt_, lane, vo_lane, vo_q); break; \
                    case 8: nsf_out_partials<8>(rs, f3_, b3_, nT, D, g0_, g1_, g2_, g3_, H2, pt_, lane, vo_lane, vo_q); break; \
                    default:                                                                                      \
                        nsf_out_partials_wide(rs, f3_, b3_, nT, D, nK, g0_, H2, pt_, lane, vo_lane, vo_q);        \
                        nsf_out_partials_wide(rs, f3_, b3_, nT, D, nK, g1_, H2, pt_ + 512, lane, vo_lane, vo_q);  \
                        nsf_out_partials_wide(rs, f3_, b3_, nT, D, nK, g2_, H2, pt_ + 1024, lane, vo_lane, vo_q); \
                        nsf_out_partials_wide(rs, f3_, b3_, nT, D, nK, g3_, H2, pt_ + 1536, lane, vo_lane, vo_q); \
                        break;                                                                                    \
                }                                                                                                 \
                lds_bar();                                                /* E(T1 - 1) */                         \
            }
            for (int T2 = 0; T2 < nTl; T2 += 2) {
                NB_TILE(T2, pA1, pA2, xA, bA0, bA1, bA2, pB1, pB2, xB, bB0, bB1, bB2)
                if (T2 + 1 >= nTl) break;
                NB_TILE(T2 + 1, pB1, pB2, xB, bB0, bB1, bB2, pA1, pA2, xA, bA0, bA1, bA2)
            }
            {   // the next transform's first operands: on their way before this one ends
                const int tbn = (t > 0 ? t - 1 : 0) * blk_bytes;
                NB_FETCH(tbn, 0, pA1, pA2, xA, bA0, bA1, bA2)
            }
            lds_bar();                                                    // E(nTl - 1)
            __syncthreads();                                              // (the chain re-ranked x)
        }
#undef NB_TILE
#undef NB_K
#undef NB_FETCH
    } else {
        // ==================================================================================== CHAIN wave
        NsfHid fA, fB;
        NsfOut ob[2];
        // the chain's hidden operands of tile U of transform tt
        auto request_hid = [&](NsfHid& F, const int tt, const int U) {
            const int base = tt * blk_bytes;
            const int Un = U + 1 < nT ? U + 1 : U;
            const int voN = U + 1 < nT ? vo_T : NSF2_OOB;
            F.wt1 = nbload4(rs, vo_T, base + oF1 + (U * nT + U) * 1024);
            F.wt2 = nbload4(rs, vo_T, base + oF2 + (U * nT + U) * 1024);
            F.wn1 = nbload4(rs, voN, base + oF1 + (Un * nT + U) * 1024);
            F.wn2 = nbload4(rs, voN, base + oF2 + (Un * nT + U) * 1024);
#pragma unroll
            for (int i = 0; i < 3; ++i) F.w0o[i] = nbload4(rs, ((4 + i) << 6) + vo_q, base + oCW0 + U * 1024);
            F.w0o[3] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 4; ++i) F.w0N[i] = nbload4(rs, U + 1 < nT ? (i << 6) + vo_q : NSF2_OOB, base + oCW0 + Un * 1024);
        };
        // rank g's two output tiles against hidden tiles Kp (previous; < 0: none) and Kc (own) of transform tt
        auto request_out = [&](NsfOut& O, const int tt, const int g, const int Kp, const int Kc) {
            const bool lv = g < D;
            const int so = tt * blk_bytes + oF3I + (lv ? g : 0) * 2 * nT * 1024;
            const int vp = (lv && Kp >= 0) ? vo_lane : NSF2_OOB, vc = lv ? vo_lane : NSF2_OOB;
            const int kp = Kp >= 0 ? Kp : 0;
            if (NSF2_ABL & 16) return;
            O.fp0 = nbload4(rs, vp, so + kp * 1024);
            O.fp1 = nbload4(rs, vp, so + (nT + kp) * 1024);
            O.fc0 = nbload4(rs, vc, so + Kc * 1024);
            O.fc1 = nbload4(rs, vc, so + (nT + Kc) * 1024);
        };
        take_table(fA, 0);
        request_hid(fA, T - 1, 0);
        request_out(ob[0], T - 1, fA.g[0], -1, 0);
        float4 w00 = nbload4(rs, vo_q, (T - 1) * blk_bytes + oCW0);      // layer 0, first tile: the column of rank 0
        float4 r00 = nbload4(rs, vo_q, (T - 1) * blk_bytes + oB3I), r01 = nbload4(rs, vo_q, (T - 1) * blk_bytes + oB3I + 64);
        for (int t = T - 1; t >= 0; --t) {
            float* X = xsel ? XB : XA;                     // zero on entry
            xsel ^= 1;
            NsfChain s;
            {   // rank 0 reads nothing: bias only
                float xv, l;
                rqs_inverse_split(as_acc(r00), as_acc(r01), PAR + (p << 5), q, Y[lidx(0, p)], xv, l);
                if (q == 0) X[lidx(0, p)] = xv;
                ladj -= l;
                WAVE_LDS_FENCE();
                s.a0N[0] = w00.x * xv; s.a0N[1] = w00.y * xv; s.a0N[2] = w00.z * xv; s.a0N[3] = w00.w * xv;
            }
            s.accN1 = s.accN2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) s.h2s[j] = 0.0f;
            lds_bar();                                                    // E(-1): the first tile's staging is complete

            for (int Tt_ = 0; Tt_ < nTl; ++Tt_) {
                const int Tt = __builtin_amdgcn_readfirstlane(Tt_);
                NsfHid& cur = fA;
                NsfHid& nxt = fB;
                const float* st = STG + (Tt & 1) * NSF2_STAGE_FLOATS;
                const float* part = PART + (Tt & 1) * NSF2_PART_FLOATS;
                const float4 s0 = *reinterpret_cast<const float4*>(st + (lane << 2));
                const float4 s1 = *reinterpret_cast<const float4*>(st + 256 + (lane << 2));
                const float4 s2 = *reinterpret_cast<const float4*>(st + 512 + (lane << 2));
                s.a0[0] = s0.x + s.a0N[0]; s.a0[1] = s0.y + s.a0N[1]; s.a0[2] = s0.z + s.a0N[2]; s.a0[3] = s0.w + s.a0N[3];
                s.p1[0] = s1.x + s.accN1[0]; s.p1[1] = s1.y + s.accN1[1]; s.p1[2] = s1.z + s.accN1[2]; s.p1[3] = s1.w + s.accN1[3];
                s.p2[0] = s2.x + s.accN2[0]; s.p2[1] = s2.y + s.accN2[1]; s.p2[2] = s2.z + s.accN2[2]; s.p2[3] = s2.w + s.accN2[3];
#pragma unroll
                for (int j = 0; j < 4; ++j) { s.a0N[j] = 0.0f; s.h2p[j] = s.h2s[j]; s.h0s[j] = s.h1s[j] = s.h2s[j] = 0.0f; }
                s.accN1 = s.accN2 = s.acc1 = s.acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
                const int pat = cur.pat;
                const int ng = __builtin_popcount(pat);      // (live groups only: the table drops the trailing padding groups)
                // what follows this tile: the next live tile, or the first tile of the next transform
                const bool more = Tt + 1 < nTl;
                const int ntt = more ? t : (t > 0 ? t - 1 : 0), nU = more ? Tt + 1 : 0;
                take_table(nxt, nU);
                auto ahead = [&](auto gi_) {
                    constexpr int G = decltype(gi_)::value;
                    if constexpr (G == 0) request_hid(nxt, ntt, nU);
                    // the next group's output fragments: a later group of this tile, or the next tile's first
                    const bool last = G + 1 >= ng;
                    const int gn = last ? nxt.g[0] : cur.g[(G + 1) & 3];
                    request_out(ob[(G + 1) & 1], last ? ntt : t, gn, last ? (more ? Tt : -1) : Tt - 1, last ? nU : Tt);
                };
                switch (pat) {
#define CASE(P) case P: nsf_group<P, 0>(s, cur, ob, part, X, Y, PAR, TAB, D, q, p, lane, ladj, ahead); break;
                    CASE(1) CASE(3) CASE(5) CASE(7) CASE(9) CASE(11) CASE(13) CASE(15)
#undef CASE
                    default: break;
                }
                {   // the tile's activations: one 16-byte word per layer and lane (row q of every quad)
                    const int hw = (Tt << 8) + (q << 6) + (p << 2);
                    *reinterpret_cast<float4*>(H0 + hw) = make_float4(s.h0s[0], s.h0s[1], s.h0s[2], s.h0s[3]);
                    *reinterpret_cast<float4*>(H1 + hw) = make_float4(s.h1s[0], s.h1s[1], s.h1s[2], s.h1s[3]);
                    *reinterpret_cast<float4*>(H2 + hw) = make_float4(s.h2s[0], s.h2s[1], s.h2s[2], s.h2s[3]);
                }
                if (ng & 1) ob[0] = ob[1];                   // (an odd number of groups leaves the next group's fragments in the second slot)
                lds_bar();                                   // E(Tt): this tile is final
                fA = fB;
            }
            w00 = nbload4(rs, vo_q, (t > 0 ? t - 1 : 0) * blk_bytes + oCW0);
            r00 = nbload4(rs, vo_q, (t > 0 ? t - 1 : 0) * blk_bytes + oB3I);
            r01 = nbload4(rs, vo_q, (t > 0 ? t - 1 : 0) * blk_bytes + oB3I + 64);
            const bool last = (t == 0);
            const int* prm = PRM + t * Dp;
            for (int e = lane; e < D * 16; e += 64) {
                const int r = e >> 4, pp = e & 15;
                const float v = X[lidx(r, pp)];
                const int tgt = prm[r];
                if (!last) Y[lidx(tgt, pp)] = v;
                else if (row0 + pp < n) out[(row0 + pp) * D + tgt] = v;
            }
            __syncthreads();
        }
        if (ladj_out && lane < 16 && row0 + p < n) ladj_out[row0 + p] = ladj;
    }
}

// Covered: spline flows whose degree groups fit a hidden tile (tri_ok), D <= 64 (NSF2_PX x tiles), one buffer resource
// over the whole image.  -1: not covered (the caller launches the lone-wave sweep).
int pmc_launch_inverse_nsf2(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n, hipStream_t stream) {
    static const int mode = getenv("PMC_INVERSE_NSF_DUO") ? atoi(getenv("PMC_INVERSE_NSF_DUO")) : -1;
    if (mode == 0) return -1;
    if (m->n_out != 23 || !m->tri_ok || m->D > 64 || m->D < 2) return -1;
    if (m->pk_per_transform * 4 * m->T >= (int64_t)NSF2_OOB) return -1;
    const size_t lds = (size_t)NSF2_LDS_FLOATS(m) * sizeof(float);
    if (lds > 160 * 1024) return -1;
    static size_t lds_set = 0;
    if (lds > 48 * 1024 && lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(maf_inverse_nsf2_kernel<0>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pmc_fail_hip(e, "hipFuncSetAttribute(maf_inverse_nsf2_kernel)");
        lds_set = lds;
    }
    hipLaunchKernelGGL(maf_inverse_nsf2_kernel<0>, dim3((unsigned)((n + 15) / 16)), dim3(128), lds, stream, *m, z, x, ladj, n);
    return pmc_check_launch("maf_inverse_nsf2_kernel");
}

// Register-resident chain of the triangular MAF inverse (used by tri4 / tri5): a tile's 16 hidden units are 4 quads
// (K = 4 of the MFMA); a degree group is one quad (or two), and the groups of a tile are run one after the other:
//   h0 = relu(a0) -> layer-1 block -> h1 -> layer-2 block -> h2 -> output rows (shift, raw) -> x -> rank-1 update of a0.
//
// TRANSPOSED ACCUMULATOR LAYOUT (round 3).  Lane (q, p) supplies row k = q of the B operand of every hop and adds the
// residual of that same row, so of a quad's four pre-activations it only ever needs "its own" row q -- but it needs that
// row for EVERY quad of the tile.  The A operand of a chain MFMA therefore carries the tile's 16 rows transposed:
// tile row i = 4*qi + r is row qi of quad r, so that accumulator register r of lane (qi, p) is row qi of quad r.  ONE
// MFMA per layer and quad (K chunk c = quad c's four activations) then adds quad c's contribution to all four quads of
// the tile at once -- the diagonal block the next hop waits for (register c) and the blocks that feed the later quads
// (registers > c; earlier quads' registers receive exact zeros: the masked weights) -- where the first register-chain
// sweeps (a quad's rows replicated 4x and rotated) needed one MFMA per (quad, later quad) pair: 12 instead of 24 chain
// MFMAs per hidden tile, 2 instead of 8 fragment loads, and no MFMA off the dependent path left on the chain wave except
// the second output accumulator.  Same products, same order of additions per pre-activation: bit-identical results.
// The output layer is not transposed (its rows are (shift, raw) pairs every lane reads by fixed index).
//
// HAND-PLACED INSTRUCTION ORDER.  The chain wave is alone on its SIMD and issues in order; scripts/micro/chain_tile.hip
// on the GPU: a hop (MFMA -> read -> add, add, max) 71 cycles, the univariate map 55, a dependent VALU operation 6, an
// LDS round trip 73 -- and every MFMA that is NOT on the dependent path still costs the wave ~36 cycles wherever it
// stands (the matrix pipe serialises them and the compiler pads result reads by instruction count).  So the groups are
// written in the order they are to be issued, with scheduling fences between the lines, side effects (stores of h, the
// previous group's x store and log-det term, work of the caller) in the shadow of the MFMA they follow, and the store
// of x unconditional (lanes that do not own the word write a scratch word: no exec-masked branch between two groups).
#ifndef PMC_MAF_CHAIN_ROT_H
#define PMC_MAF_CHAIN_ROT_H

#include <type_traits>
#include "maf_chain.h"

#define CHAIN_FENCE() __builtin_amdgcn_sched_barrier(0)

// A tile's operands of the chain wave -- static data, requested a tile ahead (two sets in rotation).
template <int MAXO>
struct ChainFrags {
    float4 wt1, wt2;               // the diagonal tile's fragments, rows transposed (chain_vo_T); component c = K chunk c
    float4 wo[2];                  // output rows (shift, raw) of the tile's groups 0, 1 / 2, 3
    float4 w0o[4];                 // [group].jt: W0[row q of quad jt of this tile][the rank the group produces]
    // right-looking mode (ABL & 2, the two-wave sweep): what this tile adds to the NEXT tile's pre-activations
    float4 wn1, wn2;               // block (Tt + 1, Tt) of layers 1 / 2, transposed
    float4 woN[2];                 // (shift, raw) rows of the next tile's groups
    float4 w0N[4];                 // [group].jt: W0[row q of quad jt of the next tile][the rank the group produces]
    float4 f3n[MAXO];              // lone-wave sweep: right-looking updates of all output tiles
    int g[4];                      // the ranks the tile's groups produce (>= D: padding group)
    int xy[4], so[4];              // two-wave sweep: byte offsets of the groups' x word (walker 0) and of their staged output partials
    int yo[4];                     // two-wave sweep: byte offsets of the groups' y word in the PREVIOUS transform's x array (no re-ranking between transforms)
    int pat;                       // two-wave sweep: the tile's quad pattern
};

template <int MAXO>
struct ChainRot {
    float a0[4];                   // layer-0 pre-activation of row q of each quad
    float p1[4], p2[4];            // layer-1/2 partial pre-activations of row q (previous tiles), added as scalars
    f32x4 acc1, acc2;              // layer-1/2 accumulators of this tile's own contributions; [r] = row q of quad r
    f32x4 outR[2];
    f32x4 oN[MAXO];
    float2 po[4];
    float yv[4];
    float h0s[4], h1s[4], h2s[4];  // activations of the tile's quads (row q): stored as one 16-byte word per layer by the last group
    f32x4 accN1, accN2;            // right-looking: layers 1 / 2 of the next tile, transposed like acc1 / acc2
    f32x4 outN[2];                 // right-looking: (shift, raw) rows of the next tile's groups
    float a0N[4];                  // right-looking: layer 0 of the next tile's quads
    float* xa[4];                  // where group i's x goes: X[lidx(g, p)] for lane quad 0 of a live group, a scratch word otherwise
    float pend_x, pend_ls;         // the previous group's x and log-scale, stored / subtracted in the next hop's shadow
    float* pend_a;
};

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// lane offset (bytes) of the transposed gather inside a natural fragment record ([16 k-lanes][16 rows] x 16 B):
// tile row i = lane & 15 carries row (i >> 2) of quad (i & 3)
__device__ __forceinline__ int chain_vo_T(int lane) {
    const int i = lane & 15;
    return (((lane >> 4) << 4) + 4 * (i & 3) + (i >> 2)) << 4;
}

// the previous group's side effects: x into LDS (every lane stores: the ones that do not own the word write a scratch
// word of their own), log-det term
template <int MAXO>
__device__ __forceinline__ void chain_flush(ChainRot<MAXO>& s, float& ladj) {
    *s.pend_a = s.pend_x;
    ladj -= s.pend_ls;
}

struct ChainNoExtra {
    template <int I, int HOP, int NG>
    __device__ __forceinline__ void operator()(std::integral_constant<int, I>, std::integral_constant<int, HOP>, std::integral_constant<int, NG>) const {}
};

// per tile, before the first group: where the groups' x go (scratch: 64 words no one reads while the groups run), no
// pending side effect
template <int MAXO>
__device__ __forceinline__ void chain_tile_begin(ChainRot<MAXO>& s, const ChainFrags<MAXO>& f, float* X, float* scratch, int D,
                                                 int q, int p, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) s.xa[i] = (q == 0 && f.g[i] < D) ? X + lidx(f.g[i] < D ? f.g[i] : 0, p) : scratch + lane;
    s.pend_a = scratch + lane;
    s.pend_x = 0.0f;
    s.pend_ls = 0.0f;
}

// (ABL: timing-only switches of scripts/abl_tri5.sh above 0xff -- results are wrong with them)
// ABL & 2: right-looking mode -- every group also adds its share to the next tile's pre-activations (accN1, accN2, outN, a0N)
// ABL & 1: no right-looking output updates -- the caller supplies the output partials of the previous tiles itself (the
// two-wave sweep: its burst wave adds them once per tile from the h2 tile this function then stores in H2[tile parity]).
//
// Groups I .. END-1 of the tile, one after the other, as STRAIGHT-LINE code: a conditional update of an
// accumulator array costs a register copy per element on every path (SSA phi), so padding groups
// (degree >= D: zero weights, zero activations) run through the same instructions and only their
// X / ladj side effects are masked; the right-looking output updates are unconditional too (the
// fragments of output tiles whose ranks are all below g are exact zeros).
// `extra(group, hop, groups of the tile)`: caller's work for the shadow of hop 0..2 of a group (the next tile's fragment requests).
// The last group's side effects stay pending: chain_flush() after the last call of a tile.
template <int PAT, int I, int END, int MAXO, int ABL = 0, class EX = ChainNoExtra>
__device__ __forceinline__ void chain_group_rot(ChainRot<MAXO>& s, const ChainFrags<MAXO>& f, float* H0, float* H1, float* X, int Tt,
                                                int D, int nOT, int q, int p, float& ladj, float* H2 = nullptr, const EX& extra = EX{}) {
    constexpr int NG = pat_ngroups(PAT);
    if constexpr (I < NG && I < END) {
        constexpr int c0 = pat_start(PAT, I), c1 = pat_end(PAT, I);
        const bool live = f.g[I] < D;
        const int hw = (Tt << 8) + (q << 6) + (p << 2);
        std::integral_constant<int, I> gi;
        std::integral_constant<int, NG> ng;
        float h0[4], h1[4], h2[4];
        // ---------------------------------------------------------------- hop 1
#pragma unroll
        for (int c = c0; c <= c1; ++c) { h0[c] = fmaxf(s.a0[c], 0.0f); s.h0s[c] = h0[c]; }
        CHAIN_FENCE();
#pragma unroll
        for (int c = c0; c <= c1; ++c) s.acc1 = MFMA(comp(f.wt1, c), h0[c], s.acc1);
        CHAIN_FENCE();
        chain_flush(s, ladj);
        if constexpr (I == NG - 1) {                   // (the tile's quads are 16 consecutive bytes per lane)
            if (!(ABL & 0x400)) *reinterpret_cast<float4*>(H0 + hw) = make_float4(s.h0s[0], s.h0s[1], s.h0s[2], s.h0s[3]);
        }
        extra(gi, std::integral_constant<int, 0>{}, ng);
        CHAIN_FENCE();
        if constexpr ((ABL & 2) != 0) {
#pragma unroll
            for (int c = c0; c <= c1; ++c) s.accN1 = MFMA(comp(f.wn1, c), h0[c], s.accN1);
            CHAIN_FENCE();
        }
#pragma unroll
        for (int c = c0; c <= c1; ++c) { h1[c] = fmaxf((s.acc1[c] + s.p1[c]) + h0[c], 0.0f); s.h1s[c] = h1[c]; }
        CHAIN_FENCE();
        // ---------------------------------------------------------------- hop 2
#pragma unroll
        for (int c = c0; c <= c1; ++c) s.acc2 = MFMA(comp(f.wt2, c), h1[c], s.acc2);
        CHAIN_FENCE();
        if constexpr (I == NG - 1) {
            if (!(ABL & 0x400)) *reinterpret_cast<float4*>(H1 + hw) = make_float4(s.h1s[0], s.h1s[1], s.h1s[2], s.h1s[3]);
        }
        extra(gi, std::integral_constant<int, 1>{}, ng);
        CHAIN_FENCE();
        if constexpr ((ABL & 2) != 0) {
#pragma unroll
            for (int c = c0; c <= c1; ++c) s.accN2 = MFMA(comp(f.wn2, c), h1[c], s.accN2);
            CHAIN_FENCE();
        }
#pragma unroll
        for (int c = c0; c <= c1; ++c) { h2[c] = fmaxf((s.acc2[c] + s.p2[c]) + h1[c], 0.0f); s.h2s[c] = h2[c]; }
        CHAIN_FENCE();
        // ---------------------------------------------------------------- hop 3: output rows of this group
        // (ABL & 8, "half" mode of the two-wave sweep: ONE output accumulator whose rows differ between the two halves of
        //  the wavefront -- lanes q < 2 receive the (shift, raw) pairs of groups 0, 1, lanes q >= 2 those of groups 2, 3 --
        //  so a group costs one output MFMA instead of two, and its x crosses to the other half with v_permlane32_swap)
        constexpr bool HALF = (ABL & 8) != 0;
        constexpr int slot = HALF ? 0 : (I >> 1);
#pragma unroll
        for (int c = c0; c <= c1; ++c) s.outR[slot] = MFMA(comp(f.wo[slot], c), h2[c], s.outR[slot]);
        CHAIN_FENCE();
        if constexpr (I == NG - 1) {                   // (two tiles deep; the lone-wave sweep uses h2 from registers only)
            if ((ABL & 1) && !(ABL & 0x400))
                *reinterpret_cast<float4*>(H2 + ((Tt & 1) << 8) + (q << 6) + (p << 2)) = make_float4(s.h2s[0], s.h2s[1], s.h2s[2], s.h2s[3]);
        }
        extra(gi, std::integral_constant<int, 2>{}, ng);
        CHAIN_FENCE();
        // ---------------------------------------------------------------- univariate map (zuko's affine inverse)
        constexpr int pi = HALF ? (I & 1) : I;                      // half mode: po / yv hold the lane's own pair of groups
        const float raw = s.outR[slot][2 * (I & 1) + 1] + s.po[pi].y;
        const float shift = s.outR[slot][2 * (I & 1)] + s.po[pi].x;
        float ls = fast_ls(raw);
        const float ydiff = s.yv[pi] - shift;
        float xg = ydiff * fast_exp_neg(ls);
        if constexpr (HALF) {
            // the half that holds this group's rows computed x and the log-scale; the other half takes them over
            const auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(xg), __float_as_uint(xg), false, false);
            const auto sl = __builtin_amdgcn_permlane32_swap(__float_as_uint(ls), __float_as_uint(ls), false, false);
            xg = __uint_as_float(sx[I >> 1]);
            ls = __uint_as_float(sl[I >> 1]);
        }
        xg = live ? xg : 0.0f;
#pragma unroll
        for (int jt = c1 + 1; jt < 4; ++jt) s.a0[jt] = fmaf(comp(f.w0o[I], jt), xg, s.a0[jt]);
        s.pend_x = xg;
        s.pend_ls = live ? ls : 0.0f;
        s.pend_a = s.xa[I];
        CHAIN_FENCE();
        if constexpr ((ABL & 2) != 0) {
            s.a0N[0] = fmaf(f.w0N[I].x, xg, s.a0N[0]); s.a0N[1] = fmaf(f.w0N[I].y, xg, s.a0N[1]);
            s.a0N[2] = fmaf(f.w0N[I].z, xg, s.a0N[2]); s.a0N[3] = fmaf(f.w0N[I].w, xg, s.a0N[3]);
#pragma unroll
            for (int c = c0; c <= c1; ++c) s.outN[0] = MFMA(comp(f.woN[0], c), h2[c], s.outN[0]);
            if constexpr (!HALF) {
#pragma unroll
                for (int c = c0; c <= c1; ++c) s.outN[1] = MFMA(comp(f.woN[1], c), h2[c], s.outN[1]);
            }
            CHAIN_FENCE();
        }
        // ---------------------------------------------------------------- off the dependent path
        if constexpr (!HALF && slot == 0 && NG > 2) {  // groups 2, 3 read their rows from the second output accumulator
#pragma unroll
            for (int c = c0; c <= c1; ++c) s.outR[1] = MFMA(comp(f.wo[1], c), h2[c], s.outR[1]);
            CHAIN_FENCE();
        }
        if (!(ABL & 1)) {
#pragma unroll
            for (int O = 0; O < MAXO; ++O)
#pragma unroll
                for (int c = c0; c <= c1; ++c) s.oN[O] = MFMA(comp(f.f3n[O], c), h2[c], s.oN[O]);
            CHAIN_FENCE();
        }
        chain_group_rot<PAT, I + 1, END, MAXO, ABL, EX>(s, f, H0, H1, X, Tt, D, nOT, q, p, ladj, H2, extra);
    }
}

#endif

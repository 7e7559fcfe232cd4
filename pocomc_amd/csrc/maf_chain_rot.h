// Register-resident chain of the triangular MAF inverse, ROTATED R layout (used by tri4).
//
// R layout: the A operand of a chain MFMA carries the 4 rows of one quad replicated
// over the 16 tile rows, so every lane holds all 4 values of the quad for its walker.  A lane,
// however, only ever USES one of them: lane (q, p) supplies row k = q of the B operand of the next
// hop and adds the residual of that same row.  Rotating the replication -- tile row i = 4*qi + r
// carries quad row (qi + r) & 3 -- puts "its own" row into accumulator register 0 of every lane:
//   * no register select per hop, 1 add + 1 max instead of 8 + 3 cndmask;
//   * the chain state per quad and layer is one float (layer 0) or reg 0 of one accumulator;
//   * the rank-1 update of layer 0 is one FMA per later quad.
// The output layer is not rotated (its rows are (shift, raw) pairs every lane reads by fixed index).
#ifndef PMC_MAF_CHAIN_ROT_H
#define PMC_MAF_CHAIN_ROT_H

#include "maf_chain.h"

template <int MAXO>
struct ChainRot {
    float a0[4];                   // layer-0 pre-activation of row q of each quad
    float p1[4], p2[4];            // layer-1/2 partial pre-activations of row q (previous tiles), added as scalars
    f32x4 a1[4], a2[4];            // layer-1/2 accumulators of this tile's own contributions; [0] = row q
    f32x4 outR[2];
    f32x4 oN[MAXO];
    float4 wd1[4], wd2[4];         // rotated R-layout diagonal fragments
    float4 wo[2];
    float4 f3n[MAXO];
    float w0r[4][4];               // W0[rank of group i][slot 4*jt + q]
    float2 po[4];
    float yv[4];
    int g[4];
};

// ABL: 1 = no right-looking output updates -- the caller supplies the output partials of the previous tiles itself (the
// two-wave sweep: its burst wave adds them once per tile from the h2 tile this function then stores in H2[tile parity]);
// timing experiments only (results are wrong): 2 = skip the non-critical hidden updates of later quads, 64 = skip the
// x update's transcendental
//
// Groups I .. END-1 of the tile, one after the other, as STRAIGHT-LINE code: a conditional update of an
// accumulator array costs a register copy per element on every path (SSA phi), so padding groups
// (degree >= D: zero weights, zero activations) run through the same instructions and only their
// X / ladj side effects are masked; the right-looking output updates are unconditional too (the
// fragments of output tiles whose ranks are all below g are exact zeros).
template <int PAT, int I, int END, int MAXO, int ABL = 0>
__device__ __forceinline__ void chain_group_rot(ChainRot<MAXO>& s, float* H0, float* H1, float* X,
                                                int Tt, int D, int nOT, int q, int p, float& ladj, float* H2 = nullptr) {
    constexpr int NG = pat_ngroups(PAT);
    if constexpr (I < NG && I < END) {
        constexpr int c0 = pat_start(PAT, I), c1 = pat_end(PAT, I);
        const int g = s.g[I];
        const bool live = g < D;
        const int hw = (Tt << 8) + (q << 6) + (p << 2);
        float h0[4], h1[4], h2[4];
#pragma unroll
        for (int c = c0; c <= c1; ++c) { h0[c] = fmaxf(s.a0[c], 0.0f); H0[hw + c] = h0[c]; }
#pragma unroll
        for (int jt = c0; jt < ((ABL & 2) ? c1 + 1 : 4); ++jt)
#pragma unroll
            for (int c = c0; c <= c1; ++c) s.a1[jt] = MFMA(comp(s.wd1[jt], c), h0[c], s.a1[jt]);
#pragma unroll
        for (int c = c0; c <= c1; ++c) { h1[c] = fmaxf((s.a1[c][0] + s.p1[c]) + h0[c], 0.0f); H1[hw + c] = h1[c]; }
#pragma unroll
        for (int jt = c0; jt < ((ABL & 2) ? c1 + 1 : 4); ++jt)
#pragma unroll
            for (int c = c0; c <= c1; ++c) s.a2[jt] = MFMA(comp(s.wd2[jt], c), h1[c], s.a2[jt]);
#pragma unroll
        for (int c = c0; c <= c1; ++c) {
            h2[c] = fmaxf((s.a2[c][0] + s.p2[c]) + h1[c], 0.0f);
            if (ABL & 1) H2[((Tt & 1) << 8) + (q << 6) + (p << 2) + c] = h2[c];   // (two tiles deep; the lone-wave sweep uses h2 from registers only)
        }
        constexpr int slot = I >> 1;
#pragma unroll
        for (int c = c0; c <= c1; ++c) s.outR[slot] = MFMA(comp(s.wo[slot], c), h2[c], s.outR[slot]);
        if constexpr (slot == 0 && NG > 2) {
#pragma unroll
            for (int c = c0; c <= c1; ++c) s.outR[1] = MFMA(comp(s.wo[1], c), h2[c], s.outR[1]);
        }
        const float shift = s.outR[slot][2 * (I & 1)] + s.po[I].x;
        const float ls = fast_ls(s.outR[slot][2 * (I & 1) + 1] + s.po[I].y);
        float xg = (ABL & 64) ? (s.yv[I] - shift) * ls : (s.yv[I] - shift) * fast_exp_neg(ls);
        xg = live ? xg : 0.0f;
        ladj -= live ? ls : 0.0f;
        if (q == 0 && live) X[lidx(g, p)] = xg;
#pragma unroll
        for (int jt = c1 + 1; jt < 4; ++jt) s.a0[jt] = fmaf(s.w0r[I][jt], xg, s.a0[jt]);
        if (!(ABL & 1)) {
#pragma unroll
            for (int O = 0; O < MAXO; ++O)
#pragma unroll
                for (int c = c0; c <= c1; ++c) s.oN[O] = MFMA(comp(s.f3n[O], c), h2[c], s.oN[O]);
        }
        chain_group_rot<PAT, I + 1, END, MAXO, ABL>(s, H0, H1, X, Tt, D, nOT, q, p, ladj, H2);
    }
}

#endif

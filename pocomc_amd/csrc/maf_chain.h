// Register-resident chain of the triangular MAF inverse (shared by the tri3 / tri4 kernels).
// See maf_inverse_tri3.hip for the idea: chain MFMAs use an A operand whose 16 rows are the 4 rows
// of ONE quad replicated, so every lane ends up with all 4 values of the quad for its walker.
#ifndef PMC_MAF_CHAIN_H
#define PMC_MAF_CHAIN_H

#include "maf_common.h"

__device__ __forceinline__ float selq(const f32x4& v, int q) {
    return q == 0 ? v[0] : (q == 1 ? v[1] : (q == 2 ? v[2] : v[3]));
}
__device__ __forceinline__ float comp(const float4& v, int c) {   // c is a compile-time constant at every call
    return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w));
}

// x = (y - shift) * exp(-ls),  ls = raw / (1 + |raw| / |log slope|): hardware reciprocal and exp2
// (about 1 ulp each) instead of three IEEE divisions and expf -- this sits on the dependent chain of every
// rank.  The difference to the dense kernels' IEEE forms is ~1e-7 relative (tests allow 1e-5).
__device__ __forceinline__ float fast_ls(float raw) {
    return raw * __builtin_amdgcn_rcpf(1.0f + fabsf(raw) * 0.14476482730108395f);    // 1/|log(1e-3)|
}
__device__ __forceinline__ float fast_exp_neg(float ls) {
    return __builtin_amdgcn_exp2f(ls * -1.4426950408889634f);
}

constexpr int pat_ngroups(int PAT) { return ((PAT >> 0) & 1) + ((PAT >> 1) & 1) + ((PAT >> 2) & 1) + ((PAT >> 3) & 1); }
constexpr int pat_start(int PAT, int i) {      // first quad of group i
    int n = -1;
    for (int j = 0; j < 4; ++j) { if ((PAT >> j) & 1) ++n; if (n == i) return j; }
    return 4;
}
constexpr int pat_end(int PAT, int i) {        // last quad of group i
    const int s = pat_start(PAT, i);
    int e = s;
    for (int j = s + 1; j < 4; ++j) { if ((PAT >> j) & 1) break; e = j; }
    return e;
}

template <int MAXO>
struct Chain {
    f32x4 a0[4], a1[4], a2[4];     // R-layout pre-activations of the tile's 4 quads
    f32x4 outR[2];                 // R-layout (shift, raw) of group pairs (0,1) and (2,3): this tile's own part
    f32x4 oN[MAXO];                // natural-layout output accumulators, all output tiles
    float4 wd1[4], wd2[4];         // R-layout diagonal fragments (target quad jt; components = source quad)
    float4 wo[2];                  // R-layout output fragments of the two pair slots
    float4 f3n[MAXO];              // natural output fragments [O][this tile]
    float4 w0r[4][4];              // layer-0 rows of the tile's ranks: [group][target quad]
    float2 po[4];                  // (shift, raw) partial of each group's rank from previous tiles
    float yv[4];                   // y of each group's rank (this lane's walker)
    int g[4];                      // degree (= rank) of each group
};


template <int PAT, int I, int MAXO>
__device__ __forceinline__ void chain_group(Chain<MAXO>& s, float* H0, float* H1, float* H2, float* X,
                                            const float* Y, int Tt, int D, int nOT, int q, int p, float& ladj) {
    constexpr int NG = pat_ngroups(PAT);
    if constexpr (I < NG) {
        constexpr int c0 = pat_start(PAT, I), c1 = pat_end(PAT, I);
        const int g = s.g[I];
        if (g < D) {                                       // padding quads carry the sentinel degree D
            const int hw = (Tt << 8) + (q << 6) + (p << 2);
            f32x4 h0[4], h1[4], h2[4];
            float b[4];
            // ---- layer 0 -> 1
#pragma unroll
            for (int c = c0; c <= c1; ++c) {
                for (int r = 0; r < 4; ++r) h0[c][r] = fmaxf(s.a0[c][r], 0.0f);
                b[c] = selq(h0[c], q);
                H0[hw + c] = b[c];
            }
#pragma unroll
            for (int jt = c0; jt < 4; ++jt)
#pragma unroll
                for (int c = c0; c <= c1; ++c) s.a1[jt] = MFMA(comp(s.wd1[jt], c), b[c], s.a1[jt]);
            // ---- layer 1 -> 2
#pragma unroll
            for (int c = c0; c <= c1; ++c) {
                for (int r = 0; r < 4; ++r) h1[c][r] = fmaxf(s.a1[c][r] + h0[c][r], 0.0f);
                b[c] = selq(h1[c], q);
                H1[hw + c] = b[c];
            }
#pragma unroll
            for (int jt = c0; jt < 4; ++jt)
#pragma unroll
                for (int c = c0; c <= c1; ++c) s.a2[jt] = MFMA(comp(s.wd2[jt], c), b[c], s.a2[jt]);
            // ---- layer 2 -> output
#pragma unroll
            for (int c = c0; c <= c1; ++c) {
                for (int r = 0; r < 4; ++r) h2[c][r] = fmaxf(s.a2[c][r] + h1[c][r], 0.0f);
                b[c] = selq(h2[c], q);
                H2[hw + c] = b[c];
            }
            constexpr int slot = I >> 1;
#pragma unroll
            for (int c = c0; c <= c1; ++c) s.outR[slot] = MFMA(comp(s.wo[slot], c), b[c], s.outR[slot]);
            if constexpr (slot == 0 && NG > 2) {
#pragma unroll
                for (int c = c0; c <= c1; ++c) s.outR[1] = MFMA(comp(s.wo[1], c), b[c], s.outR[1]);
            }
            // ---- rank g: x = (y - shift) / exp(ls)     (every lane, for its own walker)
            const float shift = s.outR[slot][2 * (I & 1)] + s.po[I].x;
            const float ls = fast_ls(s.outR[slot][2 * (I & 1) + 1] + s.po[I].y);
            const float xg = (s.yv[I] - shift) * fast_exp_neg(ls);
            ladj -= ls;
            if (q == 0) X[lidx(g, p)] = xg;
            // ---- rank-1 update of the layer-0 pre-activations of the tile's later quads
#pragma unroll
            for (int jt = c1 + 1; jt < 4; ++jt) {
                const float4 wv = s.w0r[I][jt];
                s.a0[jt][0] += wv.x * xg; s.a0[jt][1] += wv.y * xg; s.a0[jt][2] += wv.z * xg; s.a0[jt][3] += wv.w * xg;
            }
            // ---- right-looking update of the natural output accumulators (future tiles' partials)
            const int O0 = g >> 3;
#pragma unroll
            for (int O = 0; O < MAXO; ++O) {
                if (O >= O0 && O < nOT) {
#pragma unroll
                    for (int c = c0; c <= c1; ++c) s.oN[O] = MFMA(comp(s.f3n[O], c), b[c], s.oN[O]);
                }
            }
        }
        chain_group<PAT, I + 1, MAXO>(s, H0, H1, H2, X, Y, Tt, D, nOT, q, p, ladj);
    }
}


#endif

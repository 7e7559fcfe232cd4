// Helpers shared by the chains of the triangular MAF inverse (maf_chain_rot.h: register chain of the 16-row sweeps,
// maf_inverse_tri6.hip: lane-per-walker chain): fast univariate map, quad patterns of a tile.
#ifndef PMC_MAF_CHAIN_H
#define PMC_MAF_CHAIN_H

#include "maf_common.h"

// The sweeps of this family (one / two wavefronts per 16 rows, lane-per-walker) promise bit-identical results whatever
// kernel a launch size selects: no implicit contraction -- fused multiply-adds are written out (fmaf) where wanted, so
// that two kernels compiled from the same expressions cannot round differently.
#pragma clang fp contract(off)

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
    return raw * __builtin_amdgcn_rcpf(fmaf(fabsf(raw), 0.14476482730108395f, 1.0f));    // 1/|log(1e-3)|
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

#endif

// Shared device helpers of the MAF kernels (layouts, MFMA tile step, LDS staging).
#ifndef PMC_MAF_COMMON_H
#define PMC_MAF_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pmc_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// zuko MonotonicAffineTransform: log(slope) with slope = 1e-3
#define PMC_LOG_SLOPE (-6.907755278982137f)

__device__ __forceinline__ int lidx(int r, int p) {
    return ((r >> 4) << 8) + ((r & 3) << 6) + (p << 2) + ((r >> 2) & 3);
}

__device__ __forceinline__ float sel4(const float4& v, int j) {
    return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w));
}

__device__ __forceinline__ int sel4i(const int4& v, int j) {
    return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w));
}

__device__ __forceinline__ float soft_ls(float raw) {
    return raw / (1.0f + fabsf(raw / PMC_LOG_SLOPE));
}

__device__ __forceinline__ f32x4 bias4(const float* __restrict__ b, int off) {
    const float4 v = *reinterpret_cast<const float4*>(b + off);
    f32x4 r; r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w; return r;
}

// acc += W[tile rows][K tile] * act[K tile]  : one float4 weight load + one
// ds_read_b128 feed four MFMAs.
__device__ __forceinline__ f32x4 tile_mac(f32x4 acc, const float4* __restrict__ frag,
                                          const float* act, int ktile, int lane) {
    const float4 a = frag[ktile * 64 + lane];
    const float4 b = *reinterpret_cast<const float4*>(act + (ktile << 8) + (lane << 2));
    acc = MFMA(a.x, b.x, acc);
    acc = MFMA(a.y, b.y, acc);
    acc = MFMA(a.z, b.z, acc);
    acc = MFMA(a.w, b.w, acc);
    return acc;
}


// acc += sum_{K in [K0, K1)} frag[K] . act[K]  with the weight fragments of PF K tiles in flight
// (a lone wave per SIMD has nothing else to hide the latency of a fragment load) and the LDS operand
// of the next K tile read while the four MFMAs of the current one run.  The main loop is branch
// free (loads past the end are clamped to the last tile and never used) so that the compiler keeps
// the software pipeline; two accumulators keep the matrix pipe from waiting on its own result.
template <int PF = 4>
__device__ __forceinline__ f32x4 mac_range(f32x4 acc, const float4* __restrict__ frag, const float* act,
                                           int K0, int K1, int lane) {
    const int n = K1 - K0;
    if (n <= 0) return acc;
    const float4* f = frag + (size_t)K0 * 64 + lane;
    const float* bp = act + (K0 << 8) + (lane << 2);
    float4 a[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) a[j] = f[min(j, n - 1) * 64];
    float4 b = *reinterpret_cast<const float4*>(bp);
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + PF <= n; k += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const float4 bn = *reinterpret_cast<const float4*>(bp + (min(k + j + 1, n - 1) << 8));
            const float4 aj = a[j];
            a[j] = f[min(k + j + PF, n - 1) * 64];
            if (j & 1) {
                acc1 = MFMA(aj.x, b.x, acc1); acc1 = MFMA(aj.y, b.y, acc1);
                acc1 = MFMA(aj.z, b.z, acc1); acc1 = MFMA(aj.w, b.w, acc1);
            } else {
                acc = MFMA(aj.x, b.x, acc); acc = MFMA(aj.y, b.y, acc);
                acc = MFMA(aj.z, b.z, acc); acc = MFMA(aj.w, b.w, acc);
            }
            b = bn;
        }
    }
#pragma unroll
    for (int j = 0; j < PF - 1; ++j) {
        if (k + j < n) {
            const float4 bn = *reinterpret_cast<const float4*>(bp + (min(k + j + 1, n - 1) << 8));
            const float4 aj = a[j];
            if (j & 1) {
                acc1 = MFMA(aj.x, b.x, acc1); acc1 = MFMA(aj.y, b.y, acc1);
                acc1 = MFMA(aj.z, b.z, acc1); acc1 = MFMA(aj.w, b.w, acc1);
            } else {
                acc = MFMA(aj.x, b.x, acc); acc = MFMA(aj.y, b.y, acc);
                acc = MFMA(aj.z, b.z, acc); acc = MFMA(aj.w, b.w, acc);
            }
            b = bn;
        }
    }
    return acc + acc1;
}

// store the 4 rows this lane holds of tile T into an activation array
__device__ __forceinline__ void store_rows(float* act, int T, int q, int p, const f32x4& v) {
    float* base = act + (T << 8) + (p << 2) + q;
    base[0] = v[0]; base[64] = v[1]; base[128] = v[2]; base[192] = v[3];
}

struct MafView {
    const float4* f0; const float4* f1; const float4* f2; const float4* f3;
    const float* w0n; const float* b0; const float* b1; const float* b2; const float* b3;
    const float4* f3i; const float* b3i;      // spline flows only: per-rank padded output rows (inverse sweep)
};

__device__ __forceinline__ MafView maf_view(const pmc_maf_t& m, int t) {
    const float* base = m.packed + (size_t)t * m.pk_per_transform;
    MafView v;
    const size_t sz_f0 = (size_t)m.nT * m.nXT * 256, sz_f12 = (size_t)m.nT * m.nT * 256;
    const size_t sz_f3 = (size_t)m.nOT * m.nT * 256, sz_w0n = (size_t)m.Dp * m.Hp;
    const float* p = base;
    v.f0 = reinterpret_cast<const float4*>(p); p += sz_f0;
    v.f1 = reinterpret_cast<const float4*>(p); p += sz_f12;
    v.f2 = reinterpret_cast<const float4*>(p); p += sz_f12;
    v.f3 = reinterpret_cast<const float4*>(p); p += sz_f3;
    v.w0n = p; p += sz_w0n;
    v.b0 = p; p += m.Hp;
    v.b1 = p; p += m.Hp;
    v.b2 = p; p += m.Hp;
    v.b3 = p; p += (size_t)m.nOT * 16;
    v.f3i = reinterpret_cast<const float4*>(p); p += (size_t)m.D * 2 * m.nT * 256;
    v.b3i = p;
    return v;
}

// sum a per-lane partial over the 4 quads that share a particle
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// load 16 particle rows (row-major [n][D] fp32) into an LDS rank-indexed array
__device__ __forceinline__ void load_rows(float* dst, const float* __restrict__ src, int64_t row0,
                                          int64_t n, int D, int Dp, const int* __restrict__ feat_of_rank,
                                          int lane) {
    for (int e = lane; e < Dp * 16; e += 64) {
        const int r = e >> 4, p = e & 15;      // rank-major so that LDS writes spread
        float v = 0.0f;
        if (r < D && row0 + p < n) v = src[(row0 + p) * D + feat_of_rank[r]];
        dst[lidx(r, p)] = v;
    }
}

// re-rank an LDS array for the next transform, or write it out
__device__ __forceinline__ void rerank_or_store(const float* cur, float* nxt, float* __restrict__ out,
                                                int64_t row0, int64_t n, int D, int Dp,
                                                const int* __restrict__ for_cur,
                                                const int* __restrict__ rank_next, int lane) {
    for (int e = lane; e < Dp * 16; e += 64) {
        const int r = e >> 4, p = e & 15;
        if (r < D) {
            const float v = cur[lidx(r, p)];
            const int feat = for_cur[r];
            if (rank_next) nxt[lidx(rank_next[feat], p)] = v;
            else if (row0 + p < n) out[(row0 + p) * D + feat] = v;
        }
    }
    if (rank_next) {
        for (int e = lane; e < (Dp - D) * 16; e += 64) {
            const int r = D + (e >> 4), p = e & 15;
            nxt[lidx(r, p)] = 0.0f;
        }
    }
}


// hidden layers of one transform's hyper-network for 16 walkers: xin (LDS, by rank) -> H0, H1, H2
__device__ __forceinline__ void maf_hidden_pass(const pmc_maf_t& m, const MafView& w, const float* xin,
                                                float* H0, float* H1, float* H2, int lane) {
    const int q = lane >> 4, p = lane & 15;
    const int nT = m.nT, nXT = m.nXT;
    // layer 0
    for (int T = 0; T < nT; ++T) {
        f32x4 a = bias4(w.b0, 16 * T + 4 * q);
        a = mac_range(a, w.f0 + (size_t)T * nXT * 64, xin, 0, nXT, lane);
        for (int r = 0; r < 4; ++r) a[r] = fmaxf(a[r], 0.0f);
        store_rows(H0, T, q, p, a);
    }
    __syncthreads();
    // layers 1, 2: h' = relu(h + W h + b).  Units are sorted by degree, so tile T
    // only reads tiles <= T unless a degree group is wider than a tile.
    for (int layer = 1; layer <= 2; ++layer) {
        const float* Hin = layer == 1 ? H0 : H1;
        float* Hout = layer == 1 ? H1 : H2;
        const float4* f = layer == 1 ? w.f1 : w.f2;
        const float* b = layer == 1 ? w.b1 : w.b2;
        for (int T = 0; T < nT; ++T) {
            f32x4 a = bias4(b, 16 * T + 4 * q);
            const int kend = m.tri_ok ? T + 1 : nT;
            a = mac_range(a, f + (size_t)T * nT * 64, Hin, 0, kend, lane);
            const float* hb = Hin + (T << 8) + (p << 2) + q;
            a[0] = fmaxf(a[0] + hb[0], 0.0f);
            a[1] = fmaxf(a[1] + hb[64], 0.0f);
            a[2] = fmaxf(a[2] + hb[128], 0.0f);
            a[3] = fmaxf(a[3] + hb[192], 0.0f);
            store_rows(Hout, T, q, p, a);
        }
        __syncthreads();
    }
}

// 16 bytes through a bounds-checked buffer resource: wave-constant VGPR byte offset + SGPR byte offset (a lane offset
// beyond the resource's size returns zeros: how the sweeps "request nothing" without a branch)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// Compiler-only fence between dependent LDS accesses of ONE wavefront: a wave's DS
// instructions execute in issue order, so a ds_read after a ds_write needs no s_barrier
// and no s_waitcnt -- only that the compiler keeps the program order.
#define WAVE_LDS_FENCE() asm volatile("" ::: "memory")

#endif

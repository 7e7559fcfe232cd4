// Flow.forward / Flow.log_prob (pocomc/flow.py:99-114, :134-147) of the affine flows on the bf16 matrix cores:
// v_mfma_f32_16x16x32_bf16, fp32 accumulate, fp32 master weights (the bf16 fragment image is derived from them by
// pmc_maf_pack_bf16), fp32 univariate map and log-determinant.  BASELINE config 5 names this precision ("8-layer MAF
// bf16"); it is an opt-in of Flow (precision="bf16"), the float32 kernels stay the default and the parity reference.
//
// Same structure as maf_forward_wg.hip -- a workgroup of NW wavefronts owns 16 rows, the 16-unit out tiles of a layer are
// dealt to the waves, activations sit in workgroup LDS as B operands, an LDS-only barrier separates dependent layers --
// with K tiles of 32: one 16-byte fragment load and one 16-byte LDS read feed an MFMA of 16 x 16 x 32 (the float32
// kernels need four MFMAs and the same bytes for 16 x 16 x 16), activations take half the LDS.
#include <stdlib.h>
#include "maf_wg.h"
#include "bf16.h"

namespace fbf {

// acc += sum_{K2 in [0, n)} frag[K2] . act[K2]   (frag: [K2][lane] 16-byte records in global memory, act: the same in LDS),
// PF fragments in flight, two accumulators
template <int PF>
__device__ __forceinline__ f32x4 mac_bf(f32x4 acc, const uint4* __restrict__ frag, const uint4* act, int n, int lane) {
    if (n <= 0) return acc;
    const uint4* f = frag + lane;
    const uint4* b = act + lane;
    uint4 a[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) a[j] = f[min(j, n - 1) * 64];
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + PF <= n; k += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const uint4 aj = a[j];
            a[j] = f[min(k + j + PF, n - 1) * 64];
            const uint4 bj = b[(k + j) * 64];
            if (j & 1) acc1 = mfma_bf(aj, bj, acc1); else acc = mfma_bf(aj, bj, acc);
        }
    }
#pragma unroll
    for (int j = 0; j < PF - 1; ++j) {
        if (k + j < n) {
            const uint4 bj = b[(k + j) * 64];
            if (j & 1) acc1 = mfma_bf(a[j], bj, acc1); else acc = mfma_bf(a[j], bj, acc);
        }
    }
    return acc + acc1;
}

// the 4 units 16 T + 4 q + r (r = 0..3) of row p inside an activation buffer: 8 bytes
__device__ __forceinline__ int act_off(int T, int q, int p) {           // in bf16 elements
    return (((T >> 1) * 64 + (2 * (T & 1) + (q >> 1)) * 16 + p) << 3) + 4 * (q & 1);
}
__device__ __forceinline__ void store4(unsigned short* act, int T, int q, int p, const f32x4& v) {
    uint2 w;
    w.x = (unsigned)to_bf16(v[0]) | ((unsigned)to_bf16(v[1]) << 16);
    w.y = (unsigned)to_bf16(v[2]) | ((unsigned)to_bf16(v[3]) << 16);
    *reinterpret_cast<uint2*>(act + act_off(T, q, p)) = w;
}
__device__ __forceinline__ f32x4 load4(const unsigned short* act, int T, int q, int p) {
    const uint2 w = *reinterpret_cast<const uint2*>(act + act_off(T, q, p));
    return f32x4{from_bf16((unsigned short)(w.x & 0xffff)), from_bf16((unsigned short)(w.x >> 16)),
                 from_bf16((unsigned short)(w.y & 0xffff)), from_bf16((unsigned short)(w.y >> 16))};
}
// one input value of rank r, row p
__device__ __forceinline__ int x_off(int r, int p) { return (((r >> 5) * 64 + ((r & 31) >> 3) * 16 + p) << 3) + (r & 7); }

}  // namespace fbf

template <int NW>
__global__ __launch_bounds__(64 * NW) void maf_forward_bf16_kernel(pmc_maf_t m, const unsigned short* __restrict__ img,
                                                                   int64_t img_per_transform, const float* __restrict__ in,
                                                                   float* __restrict__ out, float* __restrict__ ladj_out,
                                                                   float* __restrict__ logprob_out, int64_t n,
                                                                   const int64_t* __restrict__ idx) {
    using namespace fbf;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, p = lane & 15;
    const int64_t row0 = (int64_t)blockIdx.x * 16;
    const int D = m.D, Dp = m.Dp, Hp = m.Hp, T = m.T, nT = m.nT, nOT = m.nOT;
    const int nX2 = (Dp + 31) >> 5, nK2 = (Hp + 31) >> 5;
    const int nOeff = min(nOT, (D + 7) / 8);
    // LDS: x (float32, [rank][16]) x 2, x (bf16 B operand) x 2, three activation buffers (bf16), the reduction scratch
    float* Xf = reinterpret_cast<float*>(smem_raw);
    float* Xfn = Xf + Dp * 16;
    unsigned short* Xb = reinterpret_cast<unsigned short*>(Xfn + Dp * 16);
    unsigned short* Xbn = Xb + nX2 * 512;
    unsigned short* A = Xbn + nX2 * 512;
    unsigned short* B = A + nK2 * 512;
    unsigned short* C = B + nK2 * 512;
    float* RED = reinterpret_cast<float*>(C + nK2 * 512);
    const int* feat_of_rank = m.meta + 8;
    const int* rank_of_feat = m.meta + 8 + T * D;

    for (int e = tid; e < 2 * nX2 * 512 / 2; e += 64 * NW) reinterpret_cast<unsigned*>(Xb)[e] = 0u;     // padding ranks: zeros
    lds_barrier();
    for (int e = tid; e < D * 16; e += 64 * NW) {
        const int r = e >> 4, pp = e & 15;
        float v = 0.0f;
        if (row0 + pp < n) v = in[(idx ? idx[row0 + pp] : row0 + pp) * D + feat_of_rank[r]];
        Xf[r * 16 + pp] = v;
        Xb[x_off(r, pp)] = to_bf16(v);
    }
    // (padding units of the activation buffers are written as zeros by the tiles that hold them: relu(0 + 0))
    lds_barrier();
    float ladj = 0.0f;
    for (int t = 0; t < T; ++t) {
        const MafView w = maf_view(m, t);
        const unsigned short* g = img + (size_t)t * img_per_transform;
        const uint4* g0 = reinterpret_cast<const uint4*>(g);
        const uint4* g1 = g0 + (size_t)nT * nX2 * 64;
        const uint4* g2 = g1 + (size_t)nT * nK2 * 64;
        const uint4* g3 = g2 + (size_t)nT * nK2 * 64;
        const bool last = (t + 1 == T);
        // ---- layer 0
        for (int Tt = wv; Tt < nT; Tt += NW) {
            f32x4 a = bias4(w.b0, 16 * Tt + 4 * q);
            a = mac_bf<4>(a, g0 + (size_t)Tt * nX2 * 64, reinterpret_cast<const uint4*>(Xb), nX2, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = fmaxf(a[r], 0.0f);
            store4(A, Tt, q, p, a);
        }
        if ((nT & 1) && wv == 0) store4(A, nT, q, p, f32x4{0.f, 0.f, 0.f, 0.f});      // the odd half of the last k tile
        lds_barrier();
        // ---- layers 1, 2: h' = relu(h + W h + b); units sorted by degree: tile Tt reads tiles <= Tt
        for (int layer = 1; layer <= 2; ++layer) {
            const unsigned short* Hin = layer == 1 ? A : B;
            unsigned short* Hout = layer == 1 ? B : C;
            const uint4* gf = layer == 1 ? g1 : g2;
            const float* bb = layer == 1 ? w.b1 : w.b2;
            for (int it = 0;; ++it) {                                  // most expensive tiles first, dealt in a snake
                const int r = snake_item<NW>(wv, it);
                if (r >= nT) break;
                const int Tt = nT - 1 - r;
                f32x4 a = bias4(bb, 16 * Tt + 4 * q);
                a = mac_bf<4>(a, gf + (size_t)Tt * nK2 * 64, reinterpret_cast<const uint4*>(Hin),
                              m.tri_ok ? (Tt >> 1) + 1 : nK2, lane);
                const f32x4 h = load4(Hin, Tt, q, p);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) a[rr] = fmaxf(a[rr] + h[rr], 0.0f);
                store4(Hout, Tt, q, p, a);
            }
            if ((nT & 1) && wv == 0) store4(Hout, nT, q, p, f32x4{0.f, 0.f, 0.f, 0.f});
            lds_barrier();
        }
        // ---- output layer + univariate affine map (float32)
        for (int O = wv; O < nOeff; O += NW) {
            f32x4 o = bias4(w.b3, 16 * O + 4 * q);
            o = mac_bf<4>(o, g3 + (size_t)O * nK2 * 64, reinterpret_cast<const uint4*>(C), nK2, lane);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int rank = 8 * O + 2 * q + s;
                if (rank < D) {
                    const float shift = s ? o[2] : o[0];
                    const float ls = soft_ls(s ? o[3] : o[1]);
                    const float y = Xf[rank * 16 + p] * expf(ls) + shift;
                    const int feat = feat_of_rank[t * D + rank];
                    const int rn = last ? rank : rank_of_feat[(t + 1) * D + feat];
                    Xfn[rn * 16 + p] = y;
                    Xbn[x_off(rn, p)] = to_bf16(y);
                    if (last && out && row0 + p < n) out[(row0 + p) * D + feat] = y;
                    ladj += ls;
                }
            }
        }
        lds_barrier();
        { float* s1 = Xf; Xf = Xfn; Xfn = s1; unsigned short* s2 = Xb; Xb = Xbn; Xbn = s2; }
    }
    const float l = quad_sum(ladj);
    if (lane < 16) RED[wv * 16 + lane] = l;
    lds_barrier();
    if (wv == 0) {
        float lt = 0.0f;
#pragma unroll
        for (int k = 0; k < NW; ++k) lt += RED[16 * k + p];
        if (ladj_out && lane < 16 && row0 + p < n) ladj_out[row0 + p] = lt;
        if (logprob_out) {
            float ss = 0.0f;                                            // base N(0, I) log-density of z (flow.py:147)
            for (int r = q; r < D; r += 4) { const float z = Xf[r * 16 + p]; ss += z * z; }
            ss = quad_sum(ss);
            if (lane < 16 && row0 + p < n)
                logprob_out[row0 + p] = (-0.5f * ss - 0.9189385332046727f * (float)D) + lt;
        }
    }
}

// image[i] = bf16(flat[idx[i]]) or 0
__global__ void maf_pack_bf16_kernel(const float* __restrict__ flat, const int32_t* __restrict__ idx,
                                     unsigned short* __restrict__ img, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t j = idx[i];
        img[i] = j >= 0 ? fbf::to_bf16(flat[j]) : (unsigned short)0;
    }
}

extern "C" int pmc_maf_pack_bf16(const float* flat, const int32_t* pack_idx, uint16_t* image, int64_t n, void* stream) {
    if (!flat || !pack_idx || !image || n <= 0) return pmc_fail("pmc_maf_pack_bf16: bad argument");
    int64_t grid = (n + 255) / 256; if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(maf_pack_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, flat, pack_idx, image, n);
    return pmc_check_launch("maf_pack_bf16_kernel");
}

template <int NW>
static int launch_fwd_bf16(const pmc_maf_t* m, const uint16_t* image, int64_t per_t, const float* x, float* z, float* ladj,
                           float* log_prob, int64_t n, const int64_t* idx, hipStream_t st) {
    const int nX2 = (m->Dp + 31) / 32, nK2 = (m->Hp + 31) / 32;
    const size_t lds = (size_t)2 * m->Dp * 16 * sizeof(float) + (size_t)(2 * nX2 + 3 * nK2) * 1024 + 16 * NW * sizeof(float);
    if (lds > 160 * 1024) return pmc_fail("pmc_maf_forward_bf16: flow too wide for 160 KB of LDS");
    static size_t lds_set = 0;
    if (lds > 48 * 1024 && lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(maf_forward_bf16_kernel<NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pmc_fail_hip(e, "hipFuncSetAttribute(maf_forward_bf16_kernel)");
        lds_set = lds;
    }
    hipLaunchKernelGGL((maf_forward_bf16_kernel<NW>), dim3((unsigned)((n + 15) / 16)), dim3(64 * NW), lds, st, *m, image, per_t,
                       x, z, ladj, log_prob, n, idx);
    return pmc_check_launch("maf_forward_bf16_kernel");
}

extern "C" int pmc_maf_forward_bf16(const pmc_maf_t* m, const uint16_t* image, int64_t image_per_transform, const float* x,
                                    float* z, float* ladj, float* log_prob, int64_t n, const int64_t* idx, void* stream) {
    if (!m || !m->packed || !m->meta || !image || image_per_transform <= 0) return pmc_fail("pmc_maf_forward_bf16: null descriptor field");
    if (m->n_out != 2) return pmc_fail("pmc_maf_forward_bf16: affine flows only");
    if (n == 0) return 0;
    if (!x || n < 0) return pmc_fail("pmc_maf_forward_bf16: bad argument");
    return n <= 16 * 1024 ? launch_fwd_bf16<8>(m, image, image_per_transform, x, z, ladj, log_prob, n, idx, (hipStream_t)stream)
                          : launch_fwd_bf16<4>(m, image, image_per_transform, x, z, ladj, log_prob, n, idx, (hipStream_t)stream);
}

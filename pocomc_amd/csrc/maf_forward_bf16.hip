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

#ifndef FBF_PF
#define FBF_PF 4
#endif
namespace fbf {

// acc[s] += sum_{K2 in [0, n)} frag[K2] . act_s[K2]   (frag: [K2][lane] 16-byte records in global memory, act_s: the same in
// LDS, RS row sets `stride` uint4 apart): every fragment is loaded ONCE and multiplied with the RS row sets of the workgroup
// (with 16 rows per workgroup every workgroup streamed the whole weight image through its CU: 313 workgroups x 6 MB at
// BASELINE config 5 -- what bounded the kernel); PF fragments in flight.
template <int PF, int RS>
__device__ __forceinline__ void mac_bf(f32x4 (&acc)[RS], const uint4* __restrict__ frag, const uint4* act, int stride, int n, int lane) {
    if (n <= 0) return;
    const uint4* f = frag + lane;
    const uint4* b = act + lane;
    uint4 a[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) a[j] = f[min(j, n - 1) * 64];
    f32x4 acc1[RS];
#pragma unroll
    for (int s = 0; s < RS; ++s) acc1[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + PF <= n; k += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const uint4 aj = a[j];
            a[j] = f[min(k + j + PF, n - 1) * 64];
#pragma unroll
            for (int s = 0; s < RS; ++s) {
                const uint4 bj = b[(k + j) * 64 + s * stride];
                if (j & 1) acc1[s] = mfma_bf(aj, bj, acc1[s]); else acc[s] = mfma_bf(aj, bj, acc[s]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < PF - 1; ++j) {
        if (k + j < n) {
#pragma unroll
            for (int s = 0; s < RS; ++s) {
                const uint4 bj = b[(k + j) * 64 + s * stride];
                if (j & 1) acc1[s] = mfma_bf(a[j], bj, acc1[s]); else acc[s] = mfma_bf(a[j], bj, acc[s]);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < RS; ++s) acc[s] = acc[s] + acc1[s];
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

// RS row sets of 16 rows per workgroup (their activation buffers RS times in LDS): every weight fragment is loaded once
// per workgroup and multiplied with all of them
template <int NW, int RS>
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
    const int64_t row0 = (int64_t)blockIdx.x * 16 * RS;
    const int D = m.D, Dp = m.Dp, Hp = m.Hp, T = m.T, nT = m.nT, nOT = m.nOT;
    const int nX2 = (Dp + 31) >> 5, nK2 = (Hp + 31) >> 5;
    const int nOeff = min(nOT, (D + 7) / 8);
    // LDS, per row set: x (float32, [rank][16]) x 2, x (bf16 B operand) x 2, three activation buffers (bf16); then the
    // reduction scratch.  Set strides in elements of each array's type.
    const int sXf = Dp * 16, sXb = nX2 * 512, sA = nK2 * 512;
    float* Xf = reinterpret_cast<float*>(smem_raw);
    float* Xfn = Xf + RS * sXf;
    unsigned short* Xb = reinterpret_cast<unsigned short*>(Xfn + RS * sXf);
    unsigned short* Xbn = Xb + RS * sXb;
    unsigned short* A = Xbn + RS * sXb;
    unsigned short* B = A + RS * sA;
    unsigned short* C = B + RS * sA;
    float* RED = reinterpret_cast<float*>(C + RS * sA);
    const int* feat_of_rank = m.meta + 8;
    const int* rank_of_feat = m.meta + 8 + T * D;

    for (int e = tid; e < 2 * RS * sXb / 2; e += 64 * NW) reinterpret_cast<unsigned*>(Xb)[e] = 0u;     // padding ranks: zeros
    lds_barrier();
    for (int e = tid; e < RS * D * 16; e += 64 * NW) {
        const int st = e / (D * 16), e2 = e - st * D * 16;
        const int r = e2 >> 4, pp = e2 & 15;
        const int64_t row = row0 + 16 * st + pp;
        float v = 0.0f;
        if (row < n) v = in[(idx ? idx[row] : row) * D + feat_of_rank[r]];
        Xf[st * sXf + r * 16 + pp] = v;
        Xb[st * sXb + x_off(r, pp)] = to_bf16(v);
    }
    // (padding units of the activation buffers are written as zeros by the tiles that hold them: relu(0 + 0))
    lds_barrier();
    float ladj[RS];
#pragma unroll
    for (int st = 0; st < RS; ++st) ladj[st] = 0.0f;
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
            f32x4 a[RS];
#pragma unroll
            for (int st = 0; st < RS; ++st) a[st] = bias4(w.b0, 16 * Tt + 4 * q);
            mac_bf<FBF_PF, RS>(a, g0 + (size_t)Tt * nX2 * 64, reinterpret_cast<const uint4*>(Xb), sXb / 8, nX2, lane);
#pragma unroll
            for (int st = 0; st < RS; ++st) {
#pragma unroll
                for (int r = 0; r < 4; ++r) a[st][r] = fmaxf(a[st][r], 0.0f);
                store4(A + st * sA, Tt, q, p, a[st]);
            }
        }
        if ((nT & 1) && wv == 0) {                                                     // the odd half of the last k tile
#pragma unroll
            for (int st = 0; st < RS; ++st) store4(A + st * sA, nT, q, p, f32x4{0.f, 0.f, 0.f, 0.f});
        }
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
                f32x4 a[RS];
#pragma unroll
                for (int st = 0; st < RS; ++st) a[st] = bias4(bb, 16 * Tt + 4 * q);
                mac_bf<FBF_PF, RS>(a, gf + (size_t)Tt * nK2 * 64, reinterpret_cast<const uint4*>(Hin), sA / 8,
                              m.tri_ok ? (Tt >> 1) + 1 : nK2, lane);
#pragma unroll
                for (int st = 0; st < RS; ++st) {
                    const f32x4 h = load4(Hin + st * sA, Tt, q, p);
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) a[st][rr] = fmaxf(a[st][rr] + h[rr], 0.0f);
                    store4(Hout + st * sA, Tt, q, p, a[st]);
                }
            }
            if ((nT & 1) && wv == 0) {
#pragma unroll
                for (int st = 0; st < RS; ++st) store4(Hout + st * sA, nT, q, p, f32x4{0.f, 0.f, 0.f, 0.f});
            }
            lds_barrier();
        }
        // ---- output layer + univariate affine map (float32)
        for (int O = wv; O < nOeff; O += NW) {
            f32x4 o[RS];
#pragma unroll
            for (int st = 0; st < RS; ++st) o[st] = bias4(w.b3, 16 * O + 4 * q);
            mac_bf<FBF_PF, RS>(o, g3 + (size_t)O * nK2 * 64, reinterpret_cast<const uint4*>(C), sA / 8, nK2, lane);
#pragma unroll
            for (int st = 0; st < RS; ++st) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int rank = 8 * O + 2 * q + s2;
                    if (rank < D) {
                        const float shift = s2 ? o[st][2] : o[st][0];
                        const float ls = soft_ls(s2 ? o[st][3] : o[st][1]);
                        const float y = Xf[st * sXf + rank * 16 + p] * expf(ls) + shift;
                        const int feat = feat_of_rank[t * D + rank];
                        const int rn = last ? rank : rank_of_feat[(t + 1) * D + feat];
                        Xfn[st * sXf + rn * 16 + p] = y;
                        Xbn[st * sXb + x_off(rn, p)] = to_bf16(y);
                        const int64_t row = row0 + 16 * st + p;
                        if (last && out && row < n) out[row * D + feat] = y;
                        ladj[st] += ls;
                    }
                }
            }
        }
        lds_barrier();
        { float* s1 = Xf; Xf = Xfn; Xfn = s1; unsigned short* s2 = Xb; Xb = Xbn; Xbn = s2; }
    }
#pragma unroll
    for (int st = 0; st < RS; ++st) {
        const float l = quad_sum(ladj[st]);
        if (lane < 16) RED[(st * NW + wv) * 16 + lane] = l;
    }
    lds_barrier();
    if (wv < RS) {                                   // wave st finishes row set st
        const int st = wv;
        const int64_t row = row0 + 16 * st + p;
        float lt = 0.0f;
#pragma unroll
        for (int k = 0; k < NW; ++k) lt += RED[(st * NW + k) * 16 + p];
        if (ladj_out && lane < 16 && row < n) ladj_out[row] = lt;
        if (logprob_out) {
            float ss = 0.0f;                                            // base N(0, I) log-density of z (flow.py:147)
            for (int r = q; r < D; r += 4) { const float z = Xf[st * sXf + r * 16 + p]; ss += z * z; }
            ss = quad_sum(ss);
            if (lane < 16 && row < n)
                logprob_out[row] = (-0.5f * ss - 0.9189385332046727f * (float)D) + lt;
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

static size_t fwd_bf16_lds(const pmc_maf_t* m, int nw, int rs) {
    const int nX2 = (m->Dp + 31) / 32, nK2 = (m->Hp + 31) / 32;
    return (size_t)rs * ((size_t)2 * m->Dp * 16 * sizeof(float) + (size_t)(2 * nX2 + 3 * nK2) * 1024) + (size_t)16 * nw * rs * sizeof(float);
}

template <int NW, int RS>
static int launch_fwd_bf16(const pmc_maf_t* m, const uint16_t* image, int64_t per_t, const float* x, float* z, float* ladj,
                           float* log_prob, int64_t n, const int64_t* idx, hipStream_t st) {
    const size_t lds = fwd_bf16_lds(m, NW, RS);
    if (lds > 160 * 1024) return pmc_fail("pmc_maf_forward_bf16: flow too wide for 160 KB of LDS");
    static size_t lds_set = 0;
    if (lds > 48 * 1024 && lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(maf_forward_bf16_kernel<NW, RS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pmc_fail_hip(e, "hipFuncSetAttribute(maf_forward_bf16_kernel)");
        lds_set = lds;
    }
    hipLaunchKernelGGL((maf_forward_bf16_kernel<NW, RS>), dim3((unsigned)((n + 16 * RS - 1) / (16 * RS))), dim3(64 * NW), lds, st, *m,
                       image, per_t, x, z, ladj, log_prob, n, idx);
    return pmc_check_launch("maf_forward_bf16_kernel");
}

extern "C" int pmc_maf_forward_bf16(const pmc_maf_t* m, const uint16_t* image, int64_t image_per_transform, const float* x,
                                    float* z, float* ladj, float* log_prob, int64_t n, const int64_t* idx, void* stream) {
    if (!m || !m->packed || !m->meta || !image || image_per_transform <= 0) return pmc_fail("pmc_maf_forward_bf16: null descriptor field");
    if (m->n_out != 2) return pmc_fail("pmc_maf_forward_bf16: affine flows only");
    if (n == 0) return 0;
    if (!x || n < 0) return pmc_fail("pmc_maf_forward_bf16: bad argument");
    hipStream_t st = (hipStream_t)stream;
    // Measured at BASELINE config 5 (D = 128, 8 transforms, H = 512; scripts/time_bf16_forward.py, us per call of 5000 / 512
    // rows): 8 waves x 16 rows 198 / 168; 16 waves x 16 rows 247 / 122; 8 waves x 32 rows 203 / 205; 16 waves x 32 rows
    // 151 / 148.  The call is a chain of 4 T barrier-separated layers whose cost hardly depends on the rows (fragment
    // latency, two to five tiles per wave): sixteen waves shorten the chain, two row sets per workgroup halve the
    // fragment traffic per row once every CU has work.  PMC_FWD_BF16_RS / PMC_FWD_BF16_NW force a shape (A/B runs).
    // Round 4: a wave's first PF fragments of the NEXT layer requested across the barrier (values returned from a helper and
    // selected by pointer: 89 -> 103 VGPRs, no scratch -- round 3's attempt had them assigned under branches: 115 VGPRs and a
    // scratch round trip): 157 / 124 us against 150 / 122 -- nothing: a layer does not start with an exposed round trip, its
    // two tiles per wave stream ~270 KB through a CU that takes in ~41 B/clk (scripts/micro/load_latency.hip): 75 us of the
    // 150 are that feed, the rest barrier skew between sixteen waves.  Removed again.
    static const int f_rs = pmc_env_int("PMC_FWD_BF16_RS", 0);
    static const int f_nw = pmc_env_int("PMC_FWD_BF16_NW", 0);
    int rs = f_rs ? f_rs : (n >= 4096 ? 2 : 1);
    int nw = f_nw ? f_nw : (n <= 16 * 1024 ? (m->nT >= 16 ? 16 : 8) : 4);       // (sixteen waves need sixteen tiles a layer)
    if (rs == 2 && fwd_bf16_lds(m, nw == 16 ? 16 : 8, 2) > 160 * 1024) rs = 1;
    if (rs == 2) return nw == 16 ? launch_fwd_bf16<16, 2>(m, image, image_per_transform, x, z, ladj, log_prob, n, idx, st)
                                 : launch_fwd_bf16<8, 2>(m, image, image_per_transform, x, z, ladj, log_prob, n, idx, st);
    if (nw == 16) return launch_fwd_bf16<16, 1>(m, image, image_per_transform, x, z, ladj, log_prob, n, idx, st);
    return nw == 8 ? launch_fwd_bf16<8, 1>(m, image, image_per_transform, x, z, ladj, log_prob, n, idx, st)
                   : launch_fwd_bf16<4, 1>(m, image, image_per_transform, x, z, ladj, log_prob, n, idx, st);
}

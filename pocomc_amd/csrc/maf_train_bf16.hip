// Flow.fit's loss + parameter gradient (pocomc/flow.py:297-323) for the WIDE affine flows on the bf16 matrix cores
// (BASELINE config 5: D = 128, 8 transforms, H = 512 -- "MFMA-bound flow training").
//
// Why not the structure of maf_train.hip (a workgroup per 16 rows for the whole chain): a batch is <= 512 rows
// (sampler.py:289), i.e. 32 workgroups on 256 CUs, each of which streams the whole 23 MB of fp32 weights through one
// compute unit: 1.04 ms per batch at config 5 (scripts/time_fit.py; 1.65 ms before maf_train.hip's weight gradients moved to
// their own kernel in round 6).  Here a layer is what it is, a dense product
//     Y[out][row] = W[out][in] . H[in][row]          (forward and data gradients:  A = weights, B = activations)
//     dW[out][in] = dA[out][row] . H[in][row]        (weight gradients:  both operands are activations, k = row)
// on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  Both MFMA operands want their k index contiguous (8 bf16 = one
// 16-byte load per lane), so every activation and activation gradient is stored twice -- [row][unit] and [unit][row]
// -- and the weights as W and W^T, all row-major bf16 with zeros where masked or padded (no fragment packing: a 16-byte
// load per lane reads rows of 64 contiguous bytes).  The work items of a product are its 32 x 32 output tiles: a
// workgroup per tile, its four wavefronts split the contraction (<= 5 k-steps of 32 each at H = 512: every load of the
// tile is in flight at once -- the products are L2-latency bound, not bandwidth or MFMA bound), add their partial tiles
// through LDS and each finish one 16 x 16 quarter with the layer's epilogue (bias, residual, relu / relu gate, the
// univariate map, the scatter into the canonical gradient).  Dependent layers are separate launches on one stream;
// weight-gradient products ride along in the launch of the next data-gradient product.  Per batch of <= 512 rows:
// 4 T + 2 forward and 4 T + 2 backward launches.
//
// fp32: master parameters (the bf16 image is re-derived after every optimizer step), x_t, log-scales, the loss and
// dL/dx_t; bf16: weights, hidden activations and their gradients.  Sums run in a fixed order: a fit is reproducible.
#include <stdlib.h>
#include <string>
#include "bf16.h"
#include "maf_wg.h"
#include "pmc_internal.h"

#define WIDE_NB 512                 // rows per chunk (larger batches are looped over inside the launch)
#define WIDE_THREADS 256
#define WIDE_KS 5                   // k-steps of 32 a wavefront keeps in flight

typedef unsigned short u16;

struct WideDims { int D, DK, HK, OK, T, tri; };      // tri: hidden units' degree groups fit their tiles (masked blocks can be skipped)

struct WideBufs {
    float *X, *LS, *DD, *G, *C, *RL, *WS;
    u16 *XB, *XBT, *H, *HT, *DA, *DAT, *DO, *DOT;
};

__host__ __device__ inline size_t wide_carve(const WideDims& d, char* base, WideBufs* b) {
    size_t o = 0;
    const size_t NB = WIDE_NB;
    auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o += (bytes + 255) & ~(size_t)255; return p; };
    float* X = (float*)take((size_t)(d.T + 1) * NB * d.DK * 4);
    float* LS = (float*)take((size_t)d.T * NB * d.DK * 4);
    float* DD = (float*)take((size_t)d.T * NB * d.DK * 4);
    float* G = (float*)take(NB * d.DK * 4);
    float* C = (float*)take(NB * 4);
    float* RL = (float*)take(NB * 4);
    float* WS = (float*)take(256);
    u16* XB = (u16*)take((size_t)d.T * NB * d.DK * 2);
    u16* XBT = (u16*)take((size_t)d.T * d.DK * NB * 2);
    u16* H = (u16*)take((size_t)d.T * 3 * NB * d.HK * 2);
    u16* HT = (u16*)take((size_t)d.T * 3 * d.HK * NB * 2);
    // (per transform: the weight gradients of ALL transforms are formed by one launch behind the backward sweep)
    u16* DA = (u16*)take((size_t)d.T * 3 * NB * d.HK * 2);
    u16* DAT = (u16*)take((size_t)d.T * 3 * d.HK * NB * 2);
    u16* DO = (u16*)take((size_t)d.T * NB * d.OK * 2);
    u16* DOT = (u16*)take((size_t)d.T * d.OK * NB * 2);
    if (b) *b = WideBufs{X, LS, DD, G, C, RL, WS, XB, XBT, H, HT, DA, DAT, DO, DOT};
    return o;
}

static WideDims wide_dims(const pmc_maf_t& m) {
    WideDims d;
    d.D = m.D; d.T = m.T;
    d.DK = (m.D + 31) / 32 * 32;
    d.HK = (m.Hp + 31) / 32 * 32;
    d.OK = 2 * d.DK;
    d.tri = m.tri_ok;
    return d;
}

// ------------------------------------------------------------------------------------------------ tile product
// One 32 x 32 tile of  C[m][n] = sum_k A[m][k] B[n][k]  (both row-major bf16, k contiguous), k in [0, K): the four
// wavefronts of the workgroup split the k-steps, add through LDS, and wavefront w gets the 16 x 16 quarter
// (mi, ni) = (w >> 1, w & 1) as its return value: lane (g = l >> 4, p = l & 15) holds rows 16 mi + 4 g + r, column
// 16 ni + p of the tile.
__device__ __forceinline__ f32x4 tile_product(const u16* __restrict__ A, int lda, const u16* __restrict__ B, int ldb,
                                              int m0, int n0, int K, float4* red, int wv, int lane) {
    using namespace fbf;
    const int i = lane & 15, g = lane >> 4;
    const int ks = K >> 5;
    const int kb = (ks * wv) >> 2, ke = (ks * (wv + 1)) >> 2;          // this wavefront's k-steps
    const u16* pa0 = A + (size_t)(m0 + i) * lda + 8 * g;
    const u16* pa1 = pa0 + (size_t)16 * lda;
    const u16* pb0 = B + (size_t)(n0 + i) * ldb + 8 * g;
    const u16* pb1 = pb0 + (size_t)16 * ldb;
    f32x4 c00 = {0.f, 0.f, 0.f, 0.f}, c01 = c00, c10 = c00, c11 = c00;
    for (int k0 = kb; k0 < ke; k0 += WIDE_KS) {
        uint4 a0[WIDE_KS], a1[WIDE_KS], b0[WIDE_KS], b1[WIDE_KS];
#pragma unroll
        for (int j = 0; j < WIDE_KS; ++j) {
            const int k = (k0 + j < ke ? k0 + j : ke - 1) << 5;
            a0[j] = *reinterpret_cast<const uint4*>(pa0 + k);
            a1[j] = *reinterpret_cast<const uint4*>(pa1 + k);
            b0[j] = *reinterpret_cast<const uint4*>(pb0 + k);
            b1[j] = *reinterpret_cast<const uint4*>(pb1 + k);
        }
#pragma unroll
        for (int j = 0; j < WIDE_KS; ++j) {
            if (k0 + j < ke) {
                c00 = mfma_bf(a0[j], b0[j], c00); c01 = mfma_bf(a0[j], b1[j], c01);
                c10 = mfma_bf(a1[j], b0[j], c10); c11 = mfma_bf(a1[j], b1[j], c11);
            }
        }
    }
    __syncthreads();                                             // (the previous tile's partials have been read)
    float4* mine = red + (size_t)wv * 256 + lane;
    mine[0] = make_float4(c00[0], c00[1], c00[2], c00[3]);
    mine[64] = make_float4(c01[0], c01[1], c01[2], c01[3]);
    mine[128] = make_float4(c10[0], c10[1], c10[2], c10[3]);
    mine[192] = make_float4(c11[0], c11[1], c11[2], c11[3]);
    __syncthreads();
    const float4* q = red + (size_t)wv * 64 + lane;              // quarter wv of every wavefront's partial tile
    f32x4 v;
    {
        const float4 p0 = q[0], p1 = q[256], p2 = q[512], p3 = q[768];
        v[0] = (p0.x + p1.x) + (p2.x + p3.x); v[1] = (p0.y + p1.y) + (p2.y + p3.y);
        v[2] = (p0.z + p1.z) + (p2.z + p3.z); v[3] = (p0.w + p1.w) + (p2.w + p3.w);
    }
    return v;
}

__device__ __forceinline__ void put4(u16* __restrict__ rowmajor, int ld, u16* __restrict__ transposed, int ldt, int n,
                                     int m, const f32x4& v) {
    using namespace fbf;
    const u16 h0 = to_bf16(v[0]), h1 = to_bf16(v[1]), h2 = to_bf16(v[2]), h3 = to_bf16(v[3]);
    uint2 w;
    w.x = (unsigned)h0 | ((unsigned)h1 << 16);
    w.y = (unsigned)h2 | ((unsigned)h3 << 16);
    *reinterpret_cast<uint2*>(rowmajor + (size_t)n * ld + m) = w;
    u16* t = transposed + (size_t)m * ldt + n;
    t[0] = h0; t[ldt] = h1; t[2 * (size_t)ldt] = h2; t[3 * (size_t)ldt] = h3;
}

__device__ __forceinline__ f32x4 get4(const u16* __restrict__ rowmajor, int ld, int n, int m) {
    using namespace fbf;
    const uint2 w = *reinterpret_cast<const uint2*>(rowmajor + (size_t)n * ld + m);
    return f32x4{from_bf16((u16)(w.x & 0xffff)), from_bf16((u16)(w.x >> 16)), from_bf16((u16)(w.y & 0xffff)),
                 from_bf16((u16)(w.y >> 16))};
}

struct WideView {
    const u16 *W0f, *W0b, *W1f, *W1b, *W2f, *W2b, *W3f, *W3b;
    const int *I0, *I1, *I2, *I3;
    const float *b0, *b1, *b2, *b3;
    const int *ib0, *ib1, *ib2, *ib3;
};

__device__ __forceinline__ WideView wide_view(const pmc_maf_wide_t& wd, const WideDims& d, int t) {
    WideView v;
    const size_t HK = d.HK, DK = d.DK, OK = d.OK;
    const u16* p = wd.image + (size_t)t * wd.image_per_transform;
    const int* ix = wd.image_idx + (size_t)t * wd.image_per_transform;
    size_t o = 0;
    v.W0f = p + o; v.I0 = ix + o; o += HK * DK;
    v.W0b = p + o; o += DK * HK;
    v.W1f = p + o; v.I1 = ix + o; o += HK * HK;
    v.W1b = p + o; o += HK * HK;
    v.W2f = p + o; v.I2 = ix + o; o += HK * HK;
    v.W2b = p + o; o += HK * HK;
    v.W3f = p + o; v.I3 = ix + o; o += OK * HK;
    v.W3b = p + o;
    const float* b = wd.bias + (size_t)t * wd.bias_per_transform;
    const int* ib = wd.bias_idx + (size_t)t * wd.bias_per_transform;
    v.b0 = b; v.ib0 = ib; v.b1 = b + HK; v.ib1 = ib + HK; v.b2 = b + 2 * HK; v.ib2 = ib + 2 * HK;
    v.b3 = b + 3 * HK; v.ib3 = ib + 3 * HK;
    return v;
}

// gradient of the hyper-network's outputs of transform t at (row n, feature f) from dL/dy (flow.py:309-312 through the
// affine map y = x e^{ls} + shift, ls = soft-clipped raw): d shift = gy, d raw = (gy x e^{ls} - c_n) / den^2
__device__ __forceinline__ unsigned emit_do_vals(const WideBufs& b, const WideDims& d, int t, int n, int f, float gy, float c,
                                                 float xv, float ls, float dd) {
    using namespace fbf;
    float gs = 0.0f, gr = 0.0f;
    if (f < d.D) {
        gs = gy;
        gr = (gy * xv * expf(ls) - c) * dd;
    }
    const u16 hs = to_bf16(gs), hr = to_bf16(gr);
    u16* DOt = b.DO + (size_t)t * WIDE_NB * d.OK;
    u16* DOTt = b.DOT + (size_t)t * d.OK * WIDE_NB;
    *reinterpret_cast<unsigned*>(DOt + (size_t)n * d.OK + 2 * f) = (unsigned)hs | ((unsigned)hr << 16);
    DOTt[(size_t)(2 * f) * WIDE_NB + n] = hs;
    DOTt[(size_t)(2 * f + 1) * WIDE_NB + n] = hr;
    return (unsigned)hs | ((unsigned)hr << 16);
}
__device__ __forceinline__ void emit_do(const WideBufs& b, const WideDims& d, int t, int n, int f, float gy, float c) {
    const size_t e = ((size_t)t * WIDE_NB + n) * d.DK + f;
    (void)emit_do_vals(b, d, t, n, f, gy, c, b.X[e], b.LS[e], b.DD[e]);
}

// bias gradient: rows [r0, r0 + 16) of a [unit][row] bf16 array summed over the batch rows, into the canonical gradient
__device__ __forceinline__ void bias_rows(const u16* __restrict__ AT, int r0, int NBc, const int* __restrict__ bidx,
                                          float* __restrict__ grad, bool first, int wv, int lane) {
    using namespace fbf;
    for (int r = r0 + wv; r < r0 + 16; r += 4) {
        const int gi = bidx[r];
        if (gi < 0) continue;                                     // (uniform over the wavefront)
        float s = 0.0f;
        if (8 * lane < NBc) {
            const uint4 w = *reinterpret_cast<const uint4*>(AT + (size_t)r * WIDE_NB + 8 * lane);
            const unsigned u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) s += from_bf16((u16)(u[j] & 0xffff)) + from_bf16((u16)(u[j] >> 16));
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) grad[gi] = first ? s : grad[gi] + s;
    }
}

// weight-gradient tile into the canonical gradient through the image's index map
struct DwIdx { int gi[4]; };
__device__ __forceinline__ DwIdx dw_index(const int* __restrict__ imap, int ldi, int mrow, int col) {
    DwIdx x;
#pragma unroll
    for (int r = 0; r < 4; ++r) x.gi[r] = imap[(size_t)(mrow + r) * ldi + col];
    return x;
}
__device__ __forceinline__ void scatter_dw(const DwIdx& x, const f32x4& v, float* __restrict__ grad, bool first) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (x.gi[r] >= 0) grad[x.gi[r]] = first ? v[r] : grad[x.gi[r]] + v[r];
}

// One PHASE of a chunk of <= WIDE_NB rows = one launch (a dependent kernel boundary costs ~1.5 us on this machine, a
// grid-wide barrier inside a persistent launch 4-7 us at best -- MI355X_MICROARCH.md price list; the persistent form of
// this file measured 40-70 us per phase); blockIdx.x is the work item: a 32 x 32 tile of a product or 16 bias rows.
enum { PH_WSUM = 0, PH_GATHER, PH_F0, PH_F12, PH_F3, PH_Z, PH_B1, PH_B2, PH_B3, PH_B4, PH_LOSS };

struct WideArgs {
    pmc_maf_wide_t wd;
    WideDims d;
    const float* x; const float* w; const int64_t* idx;
    float wmul;
    int64_t n_rows, c0;              // rows of the call, first row of this chunk
    int nb;                          // rows of this chunk
    int first;                       // chunk 0 overwrites the gradient, later chunks add
    int t, layer;
    float* grad; float* loss;
};

template <int PH>
__global__ __launch_bounds__(WIDE_THREADS) void maf_wide_phase_kernel(WideArgs a) {
    using namespace fbf;
    __shared__ __attribute__((aligned(16))) float4 red[4 * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, p = lane & 15;
    const int mi = wv >> 1, ni = wv & 1;
    const WideDims& d = a.d;
    const int D = d.D, DK = d.DK, HK = d.HK, OK = d.OK, T = d.T;
    const int NB = WIDE_NB;
    const int nb = a.nb, NBc = (nb + 31) & ~31, nNT = NBc >> 5;
    const bool first = a.first != 0;
    const int t = a.t;
    const int64_t c0 = a.c0;
    const float* __restrict__ x = a.x;
    const float* __restrict__ w = a.w;
    const int64_t* __restrict__ idx = a.idx;
    float* __restrict__ grad = a.grad;
    WideBufs b;
    wide_carve(d, (char*)a.wd.scratch, &b);
    const int it = blockIdx.x;

    if constexpr (PH == PH_WSUM) {                                 // sum of the call's weights (flow.py:311), fixed order
        float s = 0.0f;
        for (int64_t r = tid; r < a.n_rows; r += WIDE_THREADS) s += w[idx ? idx[r] : r];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        float* rs = reinterpret_cast<float*>(red);
        if (lane == 0) rs[wv] = s;
        __syncthreads();
        if (tid == 0) b.WS[0] = (rs[0] + rs[1]) + (rs[2] + rs[3]);
    }
    if constexpr (PH == PH_GATHER) {
        const int64_t e = (int64_t)blockIdx.x * WIDE_THREADS + tid;
        if (e < (int64_t)NBc * DK) {
            const int n = (int)(e / DK), f = (int)(e - (int64_t)n * DK);
            float v = 0.0f;
            if (n < nb && f < D) v = x[(idx ? idx[c0 + n] : c0 + n) * D + f];
            b.X[(size_t)n * DK + f] = v;
            const u16 h = to_bf16(v);
            b.XB[(size_t)n * DK + f] = h;
            b.XBT[(size_t)f * NB + n] = h;
        }
    }
    if constexpr (PH == PH_F0 || PH == PH_F12 || PH == PH_F3) {
        const WideView v = wide_view(a.wd, d, t);
        const u16* XBt = b.XB + (size_t)t * NB * DK;
        u16* H0 = b.H + (size_t)(3 * t) * NB * HK;
        u16* H1 = H0 + (size_t)NB * HK;
        u16* H2 = H1 + (size_t)NB * HK;
        u16* H0T = b.HT + (size_t)(3 * t) * HK * NB;
        u16* H1T = H0T + (size_t)HK * NB;
        u16* H2T = H1T + (size_t)HK * NB;
        const int m0 = (it / nNT) << 5, n0 = (it % nNT) << 5;
        const int mr = m0 + 16 * mi + 4 * g, nc = n0 + 16 * ni + p;
        if constexpr (PH == PH_F0) {                               // h0 = relu(W0 x + b0)
            const float4 bb = *reinterpret_cast<const float4*>(v.b0 + mr);     // (epilogue operands first: in flight with the tile's)
            f32x4 c = tile_product(v.W0f, DK, XBt, DK, m0, n0, DK, red, wv, lane);
            c[0] = fmaxf(c[0] + bb.x, 0.f); c[1] = fmaxf(c[1] + bb.y, 0.f);
            c[2] = fmaxf(c[2] + bb.z, 0.f); c[3] = fmaxf(c[3] + bb.w, 0.f);
            put4(H0, HK, H0T, NB, nc, mr, c);
        }
        if constexpr (PH == PH_F12) {                              // h' = relu(h + W h + b)
            const u16* Hin = a.layer == 1 ? H0 : H1;
            u16* Hout = a.layer == 1 ? H1 : H2;
            u16* HoutT = a.layer == 1 ? H1T : H2T;
            const u16* Wf = a.layer == 1 ? v.W1f : v.W2f;
            const float* bl = a.layer == 1 ? v.b1 : v.b2;
            const float4 bb = *reinterpret_cast<const float4*>(bl + mr);
            const f32x4 h = get4(Hin, HK, nc, mr);
            f32x4 c = tile_product(Wf, HK, Hin, HK, m0, n0, HK, red, wv, lane);
            c[0] = fmaxf((c[0] + bb.x) + h[0], 0.f); c[1] = fmaxf((c[1] + bb.y) + h[1], 0.f);
            c[2] = fmaxf((c[2] + bb.z) + h[2], 0.f); c[3] = fmaxf((c[3] + bb.w) + h[3], 0.f);
            put4(Hout, HK, HoutT, NB, nc, mr, c);
        }
        if constexpr (PH == PH_F3) {                               // output layer + univariate affine map (fp32)
            const float4 bb = *reinterpret_cast<const float4*>(v.b3 + mr);
            const int f0 = mr >> 1;
            const float2 xin = *reinterpret_cast<const float2*>(b.X + ((size_t)t * NB + nc) * DK + f0);
            const f32x4 c = tile_product(v.W3f, HK, H2, HK, m0, n0, HK, red, wv, lane);
            float y[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int f = f0 + s;
                const size_t e = ((size_t)t * NB + nc) * DK + f;
                float ls = 0.0f, dd = 0.0f;
                y[s] = 0.0f;
                if (f < D) {
                    const float shift = s ? c[2] + bb.z : c[0] + bb.x;
                    const float raw = s ? c[3] + bb.w : c[1] + bb.y;
                    const float den = 1.0f + fabsf(raw / PMC_LOG_SLOPE);
                    ls = raw / den;
                    dd = 1.0f / (den * den);
                    y[s] = (s ? xin.y : xin.x) * expf(ls) + shift;
                }
                b.LS[e] = ls;
                b.DD[e] = dd;
                b.X[e + (size_t)NB * DK] = y[s];
            }
            if (t + 1 < T) {
                const u16 h0 = to_bf16(y[0]), h1 = to_bf16(y[1]);
                *reinterpret_cast<unsigned*>(b.XB + ((size_t)(t + 1) * NB + nc) * DK + f0) = (unsigned)h0 | ((unsigned)h1 << 16);
                u16* xt = b.XBT + ((size_t)(t + 1) * DK + f0) * NB + nc;
                xt[0] = h0; xt[NB] = h1;
            }
        }
    }
    if constexpr (PH == PH_Z) {                                    // loss, dL/dz, output gradients of the last transform
        const float ws = w ? (a.wd.wsum ? *a.wd.wsum : b.WS[0]) : 1.0f;
        const int n = blockIdx.x * 4 + wv;                         // a wavefront per row
        if (n < NBc) {
            float c = 0.0f;
            if (n < nb) c = w ? w[idx ? idx[c0 + n] : c0 + n] * a.wmul / ws : 1.0f;
            const float* Z = b.X + ((size_t)T * NB + n) * DK;
            float ss = 0.0f, la = 0.0f;
            for (int f = lane; f < D; f += 64) { const float z = Z[f]; ss += z * z; }
            for (int tt = 0; tt < T; ++tt)
                for (int f = lane; f < D; f += 64) la += b.LS[((size_t)tt * NB + n) * DK + f];
            for (int o = 32; o > 0; o >>= 1) { ss += __shfl_xor(ss, o); la += __shfl_xor(la, o); }
            if (lane == 0) {
                b.C[n] = c;
                b.RL[n] = (n < nb) ? -c * ((-0.5f * ss - 0.9189385332046727f * (float)D) + la) : 0.0f;
            }
            for (int f = lane; f < DK; f += 64) {
                const float gz = (f < D) ? c * Z[f] : 0.0f;
                b.G[(size_t)n * DK + f] = gz;
                emit_do(b, d, T - 1, n, f, gz, c);
            }
        }
    }
    if constexpr (PH == PH_B1 || PH == PH_B2 || PH == PH_B3 || PH == PH_B4) {
        const int tv = t >= 0 ? t : 0;
        const WideView v = wide_view(a.wd, d, tv);
        u16* H0 = b.H + (size_t)(3 * tv) * NB * HK;
        u16* H1 = H0 + (size_t)NB * HK;
        u16* H2 = H1 + (size_t)NB * HK;
        u16* H0T = b.HT + (size_t)(3 * tv) * HK * NB;
        u16* H1T = H0T + (size_t)HK * NB;
        u16* H2T = H1T + (size_t)HK * NB;
        u16* DA2 = b.DA; u16* DA1 = DA2 + (size_t)NB * HK; u16* DA0 = DA1 + (size_t)NB * HK;
        u16* DA2T = b.DAT; u16* DA1T = DA2T + (size_t)HK * NB; u16* DA0T = DA1T + (size_t)HK * NB;
        if constexpr (PH == PH_B1) {          // da2 = relu'(h2) . W3^T do   ||   dW0, db0 of transform t + 1
            const int nA = t >= 0 ? (HK >> 5) * nNT : 0;
            const int nW = t + 1 < T ? (HK >> 5) * (DK >> 5) : 0;
            const WideView vn = wide_view(a.wd, d, t + 1 < T ? t + 1 : tv);
            if (it < nA) {
                const int m0 = (it / nNT) << 5, n0 = (it % nNT) << 5;
                const int mr = m0 + 16 * mi + 4 * g, nc = n0 + 16 * ni + p;
                const f32x4 h = get4(H2, HK, nc, mr);
                f32x4 c = tile_product(v.W3b, OK, b.DO + (size_t)tv * NB * OK, OK, m0, n0, OK, red, wv, lane);
#pragma unroll
                for (int r = 0; r < 4; ++r) c[r] = h[r] > 0.f ? c[r] : 0.f;
                put4(DA2, HK, DA2T, NB, nc, mr, c);
            } else if (it < nA + nW) {
                const int j = it - nA, nKT = DK >> 5;
                const int m0 = (j / nKT) << 5, n0 = (j % nKT) << 5;
                const DwIdx gx_ = dw_index(vn.I0, DK, m0 + 16 * mi + 4 * g, n0 + 16 * ni + p);
                const f32x4 c = tile_product(DA0T, NB, b.XBT + (size_t)(t + 1) * DK * NB, NB, m0, n0, NBc, red, wv, lane);
                scatter_dw(gx_, c, grad, first);
            } else {
                bias_rows(DA0T, (it - nA - nW) << 4, NBc, vn.ib0, grad, first, wv, lane);
            }
        }
        if constexpr (PH == PH_B2) {          // da1 = relu'(h1) . (da2 + W2^T da2)   ||   dW3, db3, db2
            const int nA = (HK >> 5) * nNT, nW = (OK >> 5) * (HK >> 5), nB3 = OK >> 4;
            if (it < nA) {
                const int m0 = (it / nNT) << 5, n0 = (it % nNT) << 5;
                const int mr = m0 + 16 * mi + 4 * g, nc = n0 + 16 * ni + p;
                const f32x4 h = get4(H1, HK, nc, mr), dp = get4(DA2, HK, nc, mr);
                f32x4 c = tile_product(v.W2b, HK, DA2, HK, m0, n0, HK, red, wv, lane);
#pragma unroll
                for (int r = 0; r < 4; ++r) c[r] = h[r] > 0.f ? c[r] + dp[r] : 0.f;
                put4(DA1, HK, DA1T, NB, nc, mr, c);
            } else if (it < nA + nW) {
                const int j = it - nA, nKT = HK >> 5;
                const int m0 = (j / nKT) << 5, n0 = (j % nKT) << 5;
                const DwIdx gx_ = dw_index(v.I3, HK, m0 + 16 * mi + 4 * g, n0 + 16 * ni + p);
                const f32x4 c = tile_product(b.DOT + (size_t)tv * OK * NB, NB, H2T, NB, m0, n0, NBc, red, wv, lane);
                scatter_dw(gx_, c, grad, first);
            } else if (it < nA + nW + nB3) {
                bias_rows(b.DOT + (size_t)tv * OK * NB, (it - nA - nW) << 4, NBc, v.ib3, grad, first, wv, lane);
            } else {
                bias_rows(DA2T, (it - nA - nW - nB3) << 4, NBc, v.ib2, grad, first, wv, lane);
            }
        }
        if constexpr (PH == PH_B3) {          // da0 = relu'(h0) . (da1 + W1^T da1)   ||   dW2, db1
            const int nA = (HK >> 5) * nNT, nW = (HK >> 5) * (HK >> 5);
            if (it < nA) {
                const int m0 = (it / nNT) << 5, n0 = (it % nNT) << 5;
                const int mr = m0 + 16 * mi + 4 * g, nc = n0 + 16 * ni + p;
                const f32x4 h = get4(H0, HK, nc, mr), dp = get4(DA1, HK, nc, mr);
                f32x4 c = tile_product(v.W1b, HK, DA1, HK, m0, n0, HK, red, wv, lane);
#pragma unroll
                for (int r = 0; r < 4; ++r) c[r] = h[r] > 0.f ? c[r] + dp[r] : 0.f;
                put4(DA0, HK, DA0T, NB, nc, mr, c);
            } else if (it < nA + nW) {
                const int j = it - nA, nKT = HK >> 5;
                const int m0 = (j / nKT) << 5, n0 = (j % nKT) << 5;
                const DwIdx gx_ = dw_index(v.I2, HK, m0 + 16 * mi + 4 * g, n0 + 16 * ni + p);
                const f32x4 c = tile_product(DA2T, NB, H1T, NB, m0, n0, NBc, red, wv, lane);
                scatter_dw(gx_, c, grad, first);
            } else {
                bias_rows(DA1T, (it - nA - nW) << 4, NBc, v.ib1, grad, first, wv, lane);
            }
        }
        if constexpr (PH == PH_B4) {          // dL/dx_t = dL/dy e^{ls} + W0^T da0 (= dL/dy of transform t - 1), that
            const int nA = (DK >> 5) * nNT;   // transform's output gradients   ||   dW1
            if (it < nA) {
                const int m0 = (it / nNT) << 5, n0 = (it % nNT) << 5;
                const int mr = m0 + 16 * mi + 4 * g, nc = n0 + 16 * ni + p;
                const float cn = b.C[nc];
                const float4 gold = *reinterpret_cast<const float4*>(b.G + (size_t)nc * DK + mr);
                const float4 lsv = *reinterpret_cast<const float4*>(b.LS + ((size_t)t * NB + nc) * DK + mr);
                const size_t ep = ((size_t)(t > 0 ? t - 1 : 0) * NB + nc) * DK + mr;      // the previous transform's x, ls, 1 / den^2
                const float4 px = *reinterpret_cast<const float4*>(b.X + ep), pl = *reinterpret_cast<const float4*>(b.LS + ep),
                             pd = *reinterpret_cast<const float4*>(b.DD + ep);
                const f32x4 c = tile_product(v.W0b, HK, DA0, HK, m0, n0, HK, red, wv, lane);
                const float go[4] = {gold.x, gold.y, gold.z, gold.w}, lv[4] = {lsv.x, lsv.y, lsv.z, lsv.w};
                const float pxv[4] = {px.x, px.y, px.z, px.w}, plv[4] = {pl.x, pl.y, pl.z, pl.w}, pdv[4] = {pd.x, pd.y, pd.z, pd.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = mr + r;
                    float gx = 0.0f;
                    if (f < D) gx = c[r] + go[r] * expf(lv[r]);
                    b.G[(size_t)nc * DK + f] = gx;
                    if (t > 0) (void)emit_do_vals(b, d, t - 1, nc, f, gx, cn, pxv[r], plv[r], pdv[r]);
                }
            } else {
                const int j = it - nA, nKT = HK >> 5;
                const int m0 = (j / nKT) << 5, n0 = (j % nKT) << 5;
                const DwIdx gx_ = dw_index(v.I1, HK, m0 + 16 * mi + 4 * g, n0 + 16 * ni + p);
                const f32x4 c = tile_product(DA1T, NB, H0T, NB, m0, n0, NBc, red, wv, lane);
                scatter_dw(gx_, c, grad, first);
            }
        }
    }
    if constexpr (PH == PH_LOSS) {                                 // batch loss in a fixed order, flow.py:321
        if (wv == 0) {
            float s = 0.0f;
            for (int n = lane; n < NBc; n += 64) s += b.RL[n];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) *a.loss += s;
        }
    }
}


// (Round 4 built "fat phases" here -- a workgroup of sixteen wavefronts per 16 rows sweeping ALL layers of ALL transforms
// with the activations in LDS, three launches instead of 68 -- and measured them SLOWER: 686 / 662 us per optimizer step of
// 512 / 64 rows against 559 / 421 for the per-layer launches.  A workgroup that sweeps all layers streams the flow's whole
// weight image (26 MB forward + transposed) through ONE CU at ~41 bytes per clock: ~130 us per sweep however few rows; a
// launch per layer spreads the same bytes over all 256 CUs.  Removed in round 5; DESIGN.md appendix A has the numbers.)

// ------------------------------------------------------------------------------------------------ host side
extern "C" int64_t pmc_maf_wide_scratch_bytes(const pmc_maf_t* m) {
    if (!m) return 0;
    const WideDims d = wide_dims(*m);
    return (int64_t)wide_carve(d, nullptr, nullptr);
}

static int wide_check(const pmc_maf_t* m, const pmc_maf_wide_t* wd, const char* who) {
    if (!m || !wd || !wd->image || !wd->image_idx || !wd->bias || !wd->bias_idx || !wd->scratch)
        return pmc_fail((std::string(who) + ": incomplete descriptor").c_str());
    if (m->n_out != 2) return pmc_fail((std::string(who) + ": affine flows only").c_str());
    const WideDims d = wide_dims(*m);
    const int64_t per = 2LL * ((int64_t)d.HK * d.DK + 2LL * d.HK * d.HK + (int64_t)d.OK * d.HK);
    if (wd->image_per_transform != per || wd->bias_per_transform != 3LL * d.HK + d.OK)
        return pmc_fail((std::string(who) + ": image sizes do not match the flow").c_str());
    if (wd->scratch_bytes < (int64_t)wide_carve(d, nullptr, nullptr))
        return pmc_fail((std::string(who) + ": scratch too small").c_str());
    return 0;
}

extern "C" int pmc_maf_wide_refresh(const pmc_maf_t* m, const pmc_maf_wide_t* wd, const float* params, void* stream) {
    if (wide_check(m, wd, "pmc_maf_wide_refresh")) return 1;
    if (!params) return pmc_fail("pmc_maf_wide_refresh: bad argument");
    if (int rc = pmc_maf_pack_bf16(params, wd->image_idx, wd->image, (int64_t)m->T * wd->image_per_transform, stream)) return rc;
    return pmc_maf_pack(params, wd->bias_idx, wd->bias, (int64_t)m->T * wd->bias_per_transform, stream);
}

static int launch_wide_part(const pmc_maf_t* m, const pmc_maf_wide_t* wd, const float* x, const float* w, const int64_t* idx,
                            float wmul, float* grad, float* loss, int64_t n, hipStream_t st, bool do_wsum) {
    WideArgs a;
    a.wd = *wd; a.d = wide_dims(*m);
    a.x = x; a.w = w; a.idx = idx; a.wmul = wmul; a.n_rows = n; a.grad = grad; a.loss = loss;
    a.c0 = 0; a.nb = 0; a.first = 1; a.t = 0; a.layer = 0;
    const WideDims& d = a.d;
    const int T = d.T, mH = d.HK >> 5, mD = d.DK >> 5, mO = d.OK >> 5;
#define PHASE(PH, GRID) hipLaunchKernelGGL((maf_wide_phase_kernel<PH>), dim3((unsigned)(GRID)), dim3(WIDE_THREADS), 0, st, a)
    if (do_wsum && w && !wd->wsum) PHASE(PH_WSUM, 1);
    for (int64_t c0 = 0; c0 < n; c0 += WIDE_NB) {
        a.c0 = c0;
        a.nb = (int)((n - c0 < WIDE_NB) ? n - c0 : WIDE_NB);
        a.first = (c0 == 0);
        const int NBc = (a.nb + 31) & ~31, nNT = NBc >> 5;
        PHASE(PH_GATHER, ((int64_t)NBc * d.DK + WIDE_THREADS - 1) / WIDE_THREADS);
        for (int t = 0; t < T; ++t) {
            a.t = t;
            PHASE(PH_F0, mH * nNT);
            a.layer = 1; PHASE(PH_F12, mH * nNT);
            a.layer = 2; PHASE(PH_F12, mH * nNT);
            PHASE(PH_F3, mO * nNT);
        }
        PHASE(PH_Z, (NBc + 3) / 4);
        for (int t = T - 1; t >= 0; --t) {
            a.t = t;
            PHASE(PH_B1, mH * nNT + (t + 1 < T ? mH * mD + (d.HK >> 4) : 0));
            PHASE(PH_B2, mH * nNT + mO * mH + (d.OK >> 4) + (d.HK >> 4));
            PHASE(PH_B3, mH * nNT + mH * mH + (d.HK >> 4));
            PHASE(PH_B4, mD * nNT + mH * mH);
        }
        a.t = -1;
        PHASE(PH_B1, mH * mD + (d.HK >> 4));                       // dW0, db0 of the first transform
        PHASE(PH_LOSS, 1);
    }
#undef PHASE
    return pmc_check_launch("maf_wide_phase_kernel");
}


// (Concurrent chains -- the batch's row ranges stepped as 2 / 4 independent chains on streams of their own, gradients added
// by one kernel behind a join -- were built and measured in round 4: 562 -> 644 (2 chains) -> 1128 us (4 chains) per
// optimizer step of 512 rows.  The chain is bound by the cost of a DEPENDENT DISPATCH (~4.7 us each, in order, whatever the
// stream), not by idle CUs that a second chain could fill: more launches, more time.  Removed again.)
static int launch_wide(const pmc_maf_t* m, const pmc_maf_wide_t* wd, const float* x, const float* w, const int64_t* idx,
                       float wmul, float* grad, float* loss, int64_t n, hipStream_t st) {
    return launch_wide_part(m, wd, x, w, idx, wmul, grad, loss, n, st, true);
}

extern "C" int pmc_maf_loss_grad_bf16(const pmc_maf_t* m, const pmc_maf_wide_t* wd, const float* x, const float* w,
                                      const int64_t* idx, float wmul, float* grad, float* loss, int64_t n, void* stream) {
    if (wide_check(m, wd, "pmc_maf_loss_grad_bf16")) return 1;
    if (!x || !grad || !loss || n < 0) return pmc_fail("pmc_maf_loss_grad_bf16: bad argument");
    if (n == 0) return 0;
    return launch_wide(m, wd, x, w, idx, wmul, grad, loss, n, (hipStream_t)stream);
}

extern "C" int pmc_maf_train_epoch_bf16(const pmc_maf_t* m, const pmc_maf_wide_t* wd, pmc_adamw_t* opt, const float* x,
                                        const float* w, const int64_t* perm, int64_t n, int64_t batch_size, float* loss,
                                        float* sq_scratch, void* stream) {
    if (wide_check(m, wd, "pmc_maf_train_epoch_bf16")) return 1;
    if (!opt || !opt->params || !opt->grad || !opt->exp_avg || !opt->exp_avg_sq || opt->n_params <= 0 || !x || !loss ||
        !sq_scratch || n < 0 || batch_size < 1)
        return pmc_fail("pmc_maf_train_epoch_bf16: bad argument");
    hipStream_t st = (hipStream_t)stream;
    for (int64_t b0 = 0; b0 < n; b0 += batch_size) {
        const int64_t nb = (n - b0 < batch_size) ? n - b0 : batch_size;
        const float* xb = perm ? x : x + b0 * m->D;
        const float* wb = (w && !perm) ? w + b0 : w;
        if (launch_wide(m, wd, xb, wb, perm ? perm + b0 : nullptr, 1000.0f, opt->grad, loss, nb, st)) return 1;
        opt->step += 1;
        if (pmc_launch_clip_adamw(opt->params, opt->grad, opt->exp_avg, opt->exp_avg_sq, opt->n_params, opt->lr, opt->beta1,
                                  opt->beta2, opt->eps, opt->weight_decay, opt->max_norm, opt->step, sq_scratch, st))
            return 1;
        if (pmc_maf_wide_refresh(m, wd, opt->params, stream)) return 1;
    }
    return 0;
}

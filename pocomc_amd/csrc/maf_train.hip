// Flow training kernels (pocomc/flow.py:165-384): loss + parameter gradient of one minibatch,
// global-norm clip + AdamW.
//
//   loss = sum_n c_n * (-log_prob(x_n)),   c_n = 1                         (flow.py:309)
//                                          c_n = w_n * 1000 / sum(w_batch) (flow.py:311-312)
//
// One wavefront owns 16 rows (same layouts as the inference kernels).  Forward keeps only the
// INPUT of every transform; the backward sweep recomputes one transform's activations at a time
// (LDS holds a single transform), then
//     d(shift, raw) -> dW3, db3 -> dh2 = W3^T . -> relu' -> dW2, db2 -> dh1 = da2 + W2^T da2 -> ...
// Data-gradient products  dh = W^T da  are MFMA bursts over pre-transposed weight fragments
// (packedT); weight-gradient tiles  dW[out][in] = sum_rows da[out][row] h[in][row]  contract over
// the wave's 16 rows with four v_mfma_f32_16x16x4_f32 and are scattered into the canonical fp32
// gradient vector with hardware fp32 atomics through a host-built index map (masked weights and
// padding map to -1 and are never touched).

#include "maf_common.h"

struct TrainView {
    const float4* f0T; const float4* f1T; const float4* f2T; const float4* f3T;
    const int4* g0; const int4* g1; const int4* g2; const int4* g3;
    const int* gb0; const int* gb1; const int* gb2; const int* gb3;
};

__device__ __forceinline__ TrainView train_view(const pmc_maf_t& m, const pmc_maf_train_t& tr, int t) {
    TrainView v;
    const size_t nT = m.nT, nXT = m.nXT, nOT = m.nOT;
    const float* p = tr.packedT + (size_t)t * tr.pkT_per_transform;
    v.f0T = reinterpret_cast<const float4*>(p); p += nXT * nT * 256;
    v.f1T = reinterpret_cast<const float4*>(p); p += nT * nT * 256;
    v.f2T = reinterpret_cast<const float4*>(p); p += nT * nT * 256;
    v.f3T = reinterpret_cast<const float4*>(p);
    const int* g = tr.gmap + (size_t)t * tr.gmap_per_transform;
    v.g0 = reinterpret_cast<const int4*>(g); g += nT * nXT * 256;
    v.g1 = reinterpret_cast<const int4*>(g); g += nT * nT * 256;
    v.g2 = reinterpret_cast<const int4*>(g); g += nT * nT * 256;
    v.g3 = reinterpret_cast<const int4*>(g); g += nOT * nT * 256;
    v.gb0 = g; g += m.Hp;
    v.gb1 = g; g += m.Hp;
    v.gb2 = g; g += m.Hp;
    v.gb3 = g;
    return v;
}

__device__ __forceinline__ void scatter4(float* __restrict__ grad, const int4 idx, const f32x4& v) {
    if (idx.x >= 0) unsafeAtomicAdd(grad + idx.x, v[0]);
    if (idx.y >= 0) unsafeAtomicAdd(grad + idx.y, v[1]);
    if (idx.z >= 0) unsafeAtomicAdd(grad + idx.z, v[2]);
    if (idx.w >= 0) unsafeAtomicAdd(grad + idx.w, v[3]);
}

// dW tile: D[i][j] = sum_p a_rows[16*Ta + i][p] * b_rows[16*Tb + j][p]
__device__ __forceinline__ f32x4 outer_tile(const float* A, int Ta, const float* B, int Tb, int lane) {
    const int i = lane & 15, kq = lane >> 4;
    const int offA = (Ta << 8) + ((i & 3) << 6) + (i >> 2);
    const int offB = (Tb << 8) + ((i & 3) << 6) + (i >> 2);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int pp = (4 * c + kq) << 2;
        acc = MFMA(A[offA + pp], B[offB + pp], acc);
    }
    return acc;
}

// bias gradient of the 4 rows a lane holds: sum over the 16 walkers, one atomic per row
__device__ __forceinline__ void bias_scatter(float* __restrict__ grad, const int* __restrict__ gb, int row0,
                                             f32x4 v, int lane) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float s = v[r];
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
        if ((lane & 15) == 0) { const int idx = gb[row0 + r]; if (idx >= 0) unsafeAtomicAdd(grad + idx, s); }
    }
}

__global__ __launch_bounds__(64) void maf_lossgrad_kernel(pmc_maf_t m, pmc_maf_train_t tr,
                                                          const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ wsum, float wmul,
                                                          float* __restrict__ grad, float* __restrict__ loss,
                                                          int64_t n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int q = lane >> 4, p = lane & 15;
    const int64_t row0 = (int64_t)blockIdx.x * 16;
    const int D = m.D, Dp = m.Dp, Hp = m.Hp, Op = 2 * m.Dp, T = m.T, nT = m.nT, nXT = m.nXT, nOT = m.nOT;
    float* XT = smem;                         // [T+1][Dp*16] inputs of every transform (+ z)
    float* H0 = XT + (size_t)(T + 1) * Dp * 16;
    float* H1 = H0 + Hp * 16;
    float* H2 = H1 + Hp * 16;
    float* DA = H2 + Hp * 16;
    float* DB = DA + Hp * 16;
    float* PHI = DB + Hp * 16;                // [Op*16] (shift, raw) by packed output row
    float* GPHI = PHI + Op * 16;
    float* G = GPHI + Op * 16;                // [Dp*16] gradient wrt the transform's output, by rank
    float* GX = G + Dp * 16;
    float* CC = GX + Dp * 16;                 // [16] per-row loss coefficient
    const int* feat_of_rank = m.meta + 8;
    const int* rank_of_feat = m.meta + 8 + T * D;

    // per-row coefficient c_n
    if (lane < 16) {
        float c = 0.0f;
        if (row0 + lane < n) c = w ? w[row0 + lane] * (wmul / *wsum) : 1.0f;
        CC[lane] = c;
    }
    load_rows(XT, x, row0, n, D, Dp, feat_of_rank, lane);
    __syncthreads();

    // ------------------------------------------------------------- forward
    float ladj = 0.0f;
    for (int t = 0; t < T; ++t) {
        const MafView wv = maf_view(m, t);
        float* xin = XT + (size_t)t * Dp * 16;
        float* xout = XT + (size_t)(t + 1) * Dp * 16;
        maf_hidden_pass(m, wv, xin, H0, H1, H2, lane);
        for (int O = 0; O < nOT; ++O) {
            if (8 * O >= D) break;
            f32x4 o = bias4(wv.b3, 16 * O + 4 * q);
            o = mac_range(o, wv.f3 + (size_t)O * nT * 64, H2, 0, nT, lane);
            for (int s = 0; s < 2; ++s) {
                const int rank = 8 * O + 2 * q + s;
                if (rank < D) {
                    const float shift = s ? o[2] : o[0];
                    const float ls = soft_ls(s ? o[3] : o[1]);
                    const float y = xin[lidx(rank, p)] * expf(ls) + shift;
                    // the next transform reads its input by its own rank order
                    const int r2 = (t + 1 < T) ? rank_of_feat[(t + 1) * D + feat_of_rank[t * D + rank]] : rank;
                    xout[lidx(r2, p)] = y;
                    ladj += ls;
                }
            }
        }
        for (int e = lane; e < (Dp - D) * 16; e += 64) xout[lidx(D + (e >> 4), e & 15)] = 0.0f;
        __syncthreads();
    }
    // ---------------------------------------------------------------- loss
    {
        const float* Z = XT + (size_t)T * Dp * 16;     // rank order of the last transform
        float ss = 0.0f;
        for (int r = q; r < D; r += 4) { const float z = Z[lidx(r, p)]; ss += z * z; }
        ss = quad_sum(ss);
        const float l = quad_sum(ladj);
        const float c = CC[p];
        if (lane < 16 && row0 + p < n) {
            const float logp = (-0.5f * ss - 0.9189385332046727f * (float)D) + l;
            unsafeAtomicAdd(loss, -c * logp);
        }
        // dL/dz = c * z
        for (int e = lane; e < Dp * 16; e += 64) {
            const int r = e >> 4, pp = e & 15;
            G[lidx(r, pp)] = (r < D) ? CC[pp] * Z[lidx(r, pp)] : 0.0f;
        }
    }
    __syncthreads();

    // ------------------------------------------------------------ backward
    for (int t = T - 1; t >= 0; --t) {
        const MafView wv = maf_view(m, t);
        const TrainView tv = train_view(m, tr, t);
        const float* X = XT + (size_t)t * Dp * 16;
        // recompute this transform's activations and (shift, raw)
        maf_hidden_pass(m, wv, X, H0, H1, H2, lane);
        for (int O = 0; O < nOT; ++O) {
            f32x4 o = bias4(wv.b3, 16 * O + 4 * q);
            if (8 * O < D) o = mac_range(o, wv.f3 + (size_t)O * nT * 64, H2, 0, nT, lane);
            store_rows(PHI, O, q, p, o);
        }
        __syncthreads();
        // element-wise part: y = x e^{ls} + shift,  L += -c * sum ls
        for (int e = lane; e < Dp * 16; e += 64) {
            const int r = e >> 4, pp = e & 15;
            float gs = 0.0f, gr = 0.0f, gx = 0.0f;
            if (r < D) {
                const float xv = X[lidx(r, pp)];
                const float raw = PHI[lidx(2 * r + 1, pp)];
                const float den = 1.0f + fabsf(raw / PMC_LOG_SLOPE);
                const float el = expf(raw / den);
                const float gy = G[lidx(r, pp)];
                gs = gy;
                gr = (gy * xv * el - CC[pp]) / (den * den);
                gx = gy * el;
            }
            GPHI[lidx(2 * r, pp)] = gs;
            GPHI[lidx(2 * r + 1, pp)] = gr;
            GX[lidx(r, pp)] = gx;
        }
        __syncthreads();
        // ---- layer 3: dW3, db3, dh2 -> da2
        for (int O = 0; O < nOT; ++O) {
            if (8 * O >= D) break;
            const float* gp = GPHI + (O << 8) + (p << 2) + q;
            f32x4 gv = {gp[0], gp[64], gp[128], gp[192]};
            bias_scatter(grad, tv.gb3, 16 * O + 4 * q, gv, lane);
            for (int K = 0; K < nT; ++K)
                scatter4(grad, tv.g3[((size_t)O * nT + K) * 64 + lane], outer_tile(GPHI, O, H2, K, lane));
        }
        for (int K = 0; K < nT; ++K) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = mac_range(a, tv.f3T + (size_t)K * nOT * 64, GPHI, 0, min(nOT, (D + 7) / 8), lane);
            const float* hb = H2 + (K << 8) + (p << 2) + q;
            a[0] = hb[0] > 0.f ? a[0] : 0.f; a[1] = hb[64] > 0.f ? a[1] : 0.f;
            a[2] = hb[128] > 0.f ? a[2] : 0.f; a[3] = hb[192] > 0.f ? a[3] : 0.f;
            store_rows(DA, K, q, p, a);
            bias_scatter(grad, tv.gb2, 16 * K + 4 * q, a, lane);
        }
        __syncthreads();
        // ---- layer 2: dW2 (da2 x h1), dh1 = da2 + W2^T da2 -> da1
        for (int To = 0; To < nT; ++To) {
            const int kend = m.tri_ok ? To + 1 : nT;
            for (int Ti = 0; Ti < kend; ++Ti)
                scatter4(grad, tv.g2[((size_t)To * nT + Ti) * 64 + lane], outer_tile(DA, To, H1, Ti, lane));
        }
        for (int Ti = 0; Ti < nT; ++Ti) {
            const float* db_ = DA + (Ti << 8) + (p << 2) + q;
            f32x4 a = {db_[0], db_[64], db_[128], db_[192]};
            a = mac_range(a, tv.f2T + (size_t)Ti * nT * 64, DA, (m.tri_ok ? Ti : 0), nT, lane);
            const float* hb = H1 + (Ti << 8) + (p << 2) + q;
            a[0] = hb[0] > 0.f ? a[0] : 0.f; a[1] = hb[64] > 0.f ? a[1] : 0.f;
            a[2] = hb[128] > 0.f ? a[2] : 0.f; a[3] = hb[192] > 0.f ? a[3] : 0.f;
            store_rows(DB, Ti, q, p, a);
            bias_scatter(grad, tv.gb1, 16 * Ti + 4 * q, a, lane);
        }
        __syncthreads();
        // ---- layer 1: dW1 (da1 x h0), dh0 = da1 + W1^T da1 -> da0 (reuses DA)
        for (int To = 0; To < nT; ++To) {
            const int kend = m.tri_ok ? To + 1 : nT;
            for (int Ti = 0; Ti < kend; ++Ti)
                scatter4(grad, tv.g1[((size_t)To * nT + Ti) * 64 + lane], outer_tile(DB, To, H0, Ti, lane));
        }
        __syncthreads();
        for (int Ti = 0; Ti < nT; ++Ti) {
            const float* db_ = DB + (Ti << 8) + (p << 2) + q;
            f32x4 a = {db_[0], db_[64], db_[128], db_[192]};
            a = mac_range(a, tv.f1T + (size_t)Ti * nT * 64, DB, (m.tri_ok ? Ti : 0), nT, lane);
            const float* hb = H0 + (Ti << 8) + (p << 2) + q;
            a[0] = hb[0] > 0.f ? a[0] : 0.f; a[1] = hb[64] > 0.f ? a[1] : 0.f;
            a[2] = hb[128] > 0.f ? a[2] : 0.f; a[3] = hb[192] > 0.f ? a[3] : 0.f;
            store_rows(DA, Ti, q, p, a);
            bias_scatter(grad, tv.gb0, 16 * Ti + 4 * q, a, lane);
        }
        __syncthreads();
        // ---- layer 0: dW0 (da0 x x), dx = gx + W0^T da0
        for (int To = 0; To < nT; ++To)
            for (int Xi = 0; Xi < nXT; ++Xi)
                scatter4(grad, tv.g0[((size_t)To * nXT + Xi) * 64 + lane], outer_tile(DA, To, X, Xi, lane));
        if (t > 0) {
            for (int Xi = 0; Xi < nXT; ++Xi) {
                const float* gb_ = GX + (Xi << 8) + (p << 2) + q;
                f32x4 a = {gb_[0], gb_[64], gb_[128], gb_[192]};
                a = mac_range(a, tv.f0T + (size_t)Xi * nT * 64, DA, 0, nT, lane);
                store_rows(PHI, Xi, q, p, a);                      // PHI is free now: staging by rank of t
            }
            __syncthreads();
            // re-rank for transform t-1 (its output order)
            for (int e = lane; e < Dp * 16; e += 64) {
                const int r = e >> 4, pp = e & 15;
                if (r < D) G[lidx(rank_of_feat[(t - 1) * D + feat_of_rank[t * D + r]], pp)] = PHI[lidx(r, pp)];
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// weighted sum of -log_prob (validation loss, flow.py:336-341) and sum of weights
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void neg_weighted_sum_kernel(const float* __restrict__ logp, const float* __restrict__ w,
                                                               const float* __restrict__ wsum, float wmul,
                                                               float* __restrict__ out, int64_t n) {
    __shared__ float red[4];
    float s = 0.0f;
    const float scale = w ? wmul / *wsum : 1.0f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
        s += -(logp[e] * (w ? w[e] * scale : 1.0f));
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}

__global__ __launch_bounds__(256) void sum_kernel(const float* __restrict__ v, float* __restrict__ out, int64_t n) {
    __shared__ float red[4];
    float s = 0.0f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) s += v[e];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}

// ---------------------------------------------------------------------------
// clip_grad_norm_ (flow.py:318) + AdamW (flow.py:268, :319)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, float* __restrict__ out, int64_t n) {
    __shared__ float red[4];
    float s = 0.0f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) s += g[e] * g[e];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ mo, float* __restrict__ vo, int64_t n,
                                                    float lr, float b1, float b2, float eps, float wd, float max_norm,
                                                    float bc1, float bc2, const float* __restrict__ sqnorm) {
    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
    float coef = 1.0f;
    if (max_norm > 0.0f) {
        coef = max_norm / (sqrtf(*sqnorm) + 1e-6f);
        coef = coef > 1.0f ? 1.0f : coef;
    }
    const float step = lr / bc1, rs2 = 1.0f / sqrtf(bc2);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const float gr = g[e] * coef;
        float pv = p[e] * (1.0f - lr * wd);
        const float m1 = b1 * mo[e] + (1.0f - b1) * gr;
        const float v1 = b2 * vo[e] + (1.0f - b2) * gr * gr;
        pv -= step * m1 / (sqrtf(v1) * rs2 + eps);
        p[e] = pv; mo[e] = m1; vo[e] = v1;
    }
}

// ---------------------------------------------------------------------------
static size_t train_lds_bytes(const pmc_maf_t& m) {
    return (size_t)((m.T + 1) * m.Dp * 16 + 5 * m.Hp * 16 + 2 * 2 * m.Dp * 16 + 2 * m.Dp * 16 + 16) * sizeof(float);
}

extern "C" int pmc_maf_loss_grad(const pmc_maf_t* m, const pmc_maf_train_t* tr, const float* x, const float* w,
                                 const float* wsum, float wmul, float* grad, float* loss, int64_t n, void* stream) {
    if (!m || !tr || !tr->packedT || !tr->gmap || !x || !grad || !loss || n < 0) return pmc_fail("pmc_maf_loss_grad: bad argument");
    if (w && !wsum) return pmc_fail("pmc_maf_loss_grad: weights need their sum");
    if (n == 0) return 0;
    const size_t lds = train_lds_bytes(*m);
    if (lds > 160 * 1024) return pmc_fail("pmc_maf_loss_grad: flow too wide for the one-wave-per-16-rows training kernel");
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(maf_lossgrad_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pmc_fail_hip(e, "hipFuncSetAttribute(maf_lossgrad_kernel)");
    }
    hipLaunchKernelGGL(maf_lossgrad_kernel, dim3((unsigned)((n + 15) / 16)), dim3(64), lds, (hipStream_t)stream, *m,
                       *tr, x, w, wsum, wmul, grad, loss, n);
    return pmc_check_launch("maf_lossgrad_kernel");
}

extern "C" int pmc_neg_weighted_sum(const float* logp, const float* w, const float* wsum, float wmul, float* out,
                                    int64_t n, void* stream) {
    if (!logp || !out || n < 0 || (w && !wsum)) return pmc_fail("pmc_neg_weighted_sum: bad argument");
    if (n == 0) return 0;
    int64_t grid = (n + 255) / 256; if (grid > 256) grid = 256;
    hipLaunchKernelGGL(neg_weighted_sum_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, logp, w, wsum,
                       wmul, out, n);
    return pmc_check_launch("neg_weighted_sum_kernel");
}

extern "C" int pmc_sum_f32(const float* v, float* out, int64_t n, void* stream) {
    if (!v || !out || n < 0) return pmc_fail("pmc_sum_f32: bad argument");
    if (n == 0) return 0;
    int64_t grid = (n + 255) / 256; if (grid > 256) grid = 256;
    hipLaunchKernelGGL(sum_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, v, out, n);
    return pmc_check_launch("sum_kernel");
}

extern "C" int pmc_adamw_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                              double lr, double beta1, double beta2, double eps, double weight_decay,
                              double max_norm, int64_t step, float* sqnorm_scratch, void* stream) {
    if (!params || !grad || !exp_avg || !exp_avg_sq || !sqnorm_scratch || n <= 0 || step < 1)
        return pmc_fail("pmc_adamw_step: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(sqnorm_scratch, 0, sizeof(float), st) != hipSuccess) return pmc_fail("pmc_adamw_step: memset");
    int64_t grid = (n + 255) / 256; if (grid > 512) grid = 512;
    if (max_norm > 0.0)
        hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)grid), dim3(256), 0, st, grad, sqnorm_scratch, n);
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)grid), dim3(256), 0, st, params, grad, exp_avg, exp_avg_sq, n,
                       (float)lr, (float)beta1, (float)beta2, (float)eps, (float)weight_decay, (float)max_norm,
                       (float)bc1, (float)bc2, (const float*)sqnorm_scratch);
    return pmc_check_launch("adamw_kernel");
}

// Flow training kernels (pocomc/flow.py:165-384): loss + parameter gradient of one minibatch,
// global-norm clip + AdamW, and the per-epoch driver.
//
//   loss = sum_n c_n * (-log_prob(x_n)),   c_n = 1                         (flow.py:309)
//                                          c_n = w_n * 1000 / sum(w_batch) (flow.py:311-312)
//
// A minibatch is at most 512 rows (sampler.py:289), so the step is latency bound, not throughput
// bound.  One WORKGROUP of TRAIN_WAVES (16; 8 for narrow spline flows) wavefronts owns 16 rows: the tiles of every layer are dealt to the
// waves (cost-balanced over the triangular layers), the activations sit in workgroup LDS in
// the MFMA operand layout of the inference kernels, and a barrier separates dependent layers.
// Forward stores the input and the three hidden activations of every transform in a global scratch
// (L2 resident; without the activation scratch the backward sweep recomputes them); the backward
// sweep takes one transform at a time:
//     d(shift, raw) -> dW3, db3 -> dh2 = W3^T . -> relu' -> dW2, db2 -> dh1 = da2 + W2^T da2 -> ...
// Data-gradient products  dh = W^T da  are MFMA bursts over pre-transposed weight fragments
// (packedT); weight-gradient tiles  dW[out][in] = sum_rows da[out][row] h[in][row]  contract over
// the 16 rows with four v_mfma_f32_16x16x4_f32.
//
// No atomics: every workgroup owns a gradient SLAB in tile order (one coalesced float4 store per
// lane per tile; a workgroup that processes several row sets read-modify-writes its own slab), and
// reduce_slabs_kernel sums the slabs into the canonical gradient through the host-built map
// (masked weights and padding map to -1 and are never touched).  Sums run in a fixed order, so a
// training run is bitwise reproducible.
//
// LDS buffers alias along the backward sweep (4 hidden-width buffers instead of 7):
//     E: x_t -> da2 -> da0      A: h0      B: h1 -> x_t (reload for dW0)      C: h2 -> da1
//     P: (shift, raw) -> their gradients -> dx      G: dL/dy -> direct dL/dx term -> dL/dy of t-1
// which keeps the 8-transform, H=512, D=128 flow (BASELINE config 5) inside 160 KB.

#include <string>
#include "maf_common.h"
#include "maf_wg.h"
#include "rqs.h"

#ifndef TRAIN_PF
#define TRAIN_PF 4                  // weight fragments in flight per wave
#endif
#ifndef TRAIN_WARM_L2
#define TRAIN_WARM_L2 0      // measured neutral on MI355X (the phases are not L2-miss bound)
#endif
#ifndef TRAIN_WAVES
#define TRAIN_WAVES 16              // waves of a training workgroup ...
#endif
// ... except for narrow spline flows (hidden width <= 64): their layers have fewer tiles than that many waves, the
// spline evaluation runs on four waves either way, and a barrier over 8 waves is cheaper -- measured on MI355X per
// 256-row batch: nsf6 D=4 142 -> 113 us, D=10 157 -> 135 us, D=20 272 -> 243 us (D=32, H=128: 227 -> 256 us, so not
// there; the affine flows do not care).  With half the waves a wave has 256 registers: 8 fragments in flight.
#define TRAIN_WAVES_NARROW 8
#define TRAIN_PF_NARROW 8
static int train_waves_of(const pmc_maf_t& m) {
    return (m.n_out == RQS_NOUT && m.nT <= 6) ? TRAIN_WAVES_NARROW : TRAIN_WAVES;     // hidden width <= 64
}

struct TrainView {
    const float4* f0T; const float4* f1T; const float4* f2T; const float4* f3T;
    // slab offsets (floats) of this transform's gradient tiles / bias rows
    int64_t g0, g1, g2, g3, gb0, gb1, gb2, gb3;
};

__device__ __forceinline__ TrainView train_view(const pmc_maf_t& m, const pmc_maf_train_t& tr, int t) {
    TrainView v;
    const size_t nT = m.nT, nXT = m.nXT, nOT = m.nOT;
    const float* p = tr.packedT + (size_t)t * tr.pkT_per_transform;
    v.f0T = reinterpret_cast<const float4*>(p); p += nXT * nT * 256;
    v.f1T = reinterpret_cast<const float4*>(p); p += nT * nT * 256;
    v.f2T = reinterpret_cast<const float4*>(p); p += nT * nT * 256;
    v.f3T = reinterpret_cast<const float4*>(p);
    int64_t g = (int64_t)t * tr.gmap_per_transform;
    v.g0 = g; g += nT * nXT * 256;
    v.g1 = g; g += nT * nT * 256;
    v.g2 = g; g += nT * nT * 256;
    v.g3 = g; g += nOT * nT * 256;
    v.gb0 = g; g += m.Hp;
    v.gb1 = g; g += m.Hp;
    v.gb2 = g; g += m.Hp;
    v.gb3 = g;
    return v;
}

__device__ __forceinline__ void slab_put4(float* __restrict__ dst, const f32x4& v, bool first) {
    float4* d = reinterpret_cast<float4*>(dst);
    float4 o = make_float4(v[0], v[1], v[2], v[3]);
    if (!first) { const float4 c = *d; o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
    *d = o;
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

// bias gradient of the 4 rows a lane holds: sum over the 16 rows of the set, written by lane p == 0
__device__ __forceinline__ void bias_put(float* __restrict__ dst, f32x4 v, int lane, bool first) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float s = v[r];
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
        if ((lane & 15) == 0) dst[r] = first ? s : dst[r] + s;
    }
}

// out-layer panel of ranks [16c, 16c+16) of a spline flow: output tiles 23c .. 23c+22 -> P (local tile index).
// keep != NULL: the panel (23 tiles of 256 floats, one float4 per lane and tile) is also written there for the
// backward sweep; from != NULL: it is read back from there instead of being multiplied out.
template <int NW, int PF>
__device__ __forceinline__ void rqs_panel_train(const pmc_maf_t& m, const MafView& w, const float* H2, float* P, int c,
                                                int wv, int lane, float* keep = nullptr, const float* from = nullptr) {
    const int q = lane >> 4, p = lane & 15;
    for (int i = wv; i < RQS_NOUT; i += NW) {
        const int O = RQS_NOUT * c + i;
        if (16 * O >= RQS_NOUT * m.D) continue;              // padding rows: never read
        f32x4 o;
        if (from) {
            const float4 v = reinterpret_cast<const float4*>(from)[i * 64 + lane];
            o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        } else {
            o = bias4(w.b3, 16 * O + 4 * q);
            o = mac_range<PF>(o, w.f3 + (size_t)O * m.nT * 64, H2, 0, m.nT, lane);
            if (keep) reinterpret_cast<float4*>(keep)[i * 64 + lane] = make_float4(o[0], o[1], o[2], o[3]);
        }
        store_rows(P, i, q, p, o);
    }
}

// UNI 0: affine univariate (MAF), 2 outputs per feature.  UNI 1: 8-bin spline (NSF), 23 outputs.
template <int NW, int PF, bool PROF, int UNI>
__global__ __launch_bounds__(64 * NW) void maf_lossgrad_kernel(pmc_maf_t m, pmc_maf_train_t tr,
                                                                     const float* __restrict__ x,
                                                                     const float* __restrict__ w,
                                                                     const int64_t* __restrict__ idx, float wmul,
                                                                     int64_t n, long long* __restrict__ prof) {
    long long pacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tk = TICKT();
    const long long t_begin = tk;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: tile ownership branches stay scalar
    const int q = lane >> 4, p = lane & 15;
    const int D = m.D, Dp = m.Dp, Hp = m.Hp, T = m.T, nT = m.nT, nXT = m.nXT, nOT = m.nOT;
    const int Psz = UNI ? RQS_NOUT * 256 : 2 * Dp * 16;       // spline flows: one 16-rank panel at a time
    const int nOeff = UNI ? (RQS_NOUT * D + 15) / 16 : min(nOT, (D + 7) / 8);   // output tiles with real rows
    float* A = smem;
    float* B = A + Hp * 16;
    float* Cb = B + Hp * 16;
    float* E = Cb + Hp * 16;
    float* P = E + Hp * 16;                   // [Psz]
    float* Gb = P + Psz;                      // [Dp*16]
    float* CC = Gb + Dp * 16;                 // [16] per-row loss coefficient
    float* RED = CC + 16;                     // [16 * NW]
    float* XB = RED + 16 * NW;       // UNI 1: [Dp*16] the transform's input during its backward sweep
    const int* feat_of_rank = m.meta + 8;
    const int* rank_of_feat = m.meta + 8 + T * D;
    float* slab = tr.slabs + (size_t)blockIdx.x * tr.slab_stride;
    float* xt = tr.xt_scratch + (size_t)blockIdx.x * (T + 1) * Dp * 16;
    // hidden activations of every transform, kept from the forward sweep (the backward sweep then loads them
    // instead of recomputing three layers); NULL: recompute
    float* act = tr.act_scratch ? tr.act_scratch + (size_t)blockIdx.x * T * 3 * Hp * 16 : nullptr;
    float* par = (UNI == 1 && tr.par_scratch) ? tr.par_scratch + (size_t)blockIdx.x * T * m.nXT * RQS_NOUT * 256 : nullptr;

    // sum of the batch weights, the same fixed-order sum in every workgroup (flow.py:311)
    float wscale = 1.0f;
    if (w && tr.wsum) {
        wscale = wmul / *tr.wsum;
    } else if (w) {
        float s = 0.0f;
        for (int64_t i = tid; i < n; i += (64 * NW)) s += w[idx ? idx[i] : i];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) RED[wv] = s;
        __syncthreads();
        float tot = 0.0f;
        for (int k = 0; k < NW; ++k) tot += RED[k];
        wscale = wmul / tot;
        __syncthreads();
    }

    // Both weight images were rewritten by the previous batch's optimizer step, so this XCD's L2 holds none of
    // them and every layer would start with a miss to MALL / HBM (~1.5-2 k cycles, ~45 dependent phases).
    // Workgroups are dealt round-robin to the 8 XCDs: the workgroups of one XCD each touch a share of the two
    // images once, up front (one dword per 128-byte line, consumed at the very end so nothing waits on them).
    float warm = 0.0f;
    if (TRAIN_WARM_L2) {
        const int per_xcd = max(1, (int)((gridDim.x + 7) >> 3));
        const int share = min((int)(blockIdx.x >> 3), per_xcd - 1);
        const int64_t lines_a = ((int64_t)T * m.pk_per_transform * 4) >> 7, lines_b = ((int64_t)T * tr.pkT_per_transform * 4) >> 7;
        for (int64_t l = (int64_t)share * (64 * NW) + tid; l < lines_a + lines_b; l += (int64_t)per_xcd * (64 * NW))
            warm += (l < lines_a) ? m.packed[l << 5] : tr.packedT[(l - lines_a) << 5];
    }

    // contiguous ranges of weight-gradient tiles per phase (layer 3 | layers 2,1 | layer 0), balanced on the host
    // against the data-gradient tiles the dealing rules give each wave (MAFSpec.train_schedule)
    const int dw3_start = tr.sched[(0 * NW + wv) * 2], dw3_count = tr.sched[(0 * NW + wv) * 2 + 1];
    const int dwt_start = tr.sched[(1 * NW + wv) * 2], dwt_count = tr.sched[(1 * NW + wv) * 2 + 1];
    const int dw0_start = tr.sched[(2 * NW + wv) * 2], dw0_count = tr.sched[(2 * NW + wv) * 2 + 1];

    float loss_acc = 0.0f;
    bool first = true;
    const int64_t nsets = (n + 15) / 16;
    for (int64_t set = blockIdx.x; set < nsets; set += gridDim.x, first = false) {
        const int64_t row0 = set * 16;
        if (tid < 16) {
            float c = 0.0f;
            if (row0 + tid < n) {
                const int64_t r = idx ? idx[row0 + tid] : row0 + tid;
                c = w ? w[r] * wscale : 1.0f;
            }
            CC[tid] = c;
        }
        float* Xc = E;
        float* Xn = Gb;
        for (int e = tid; e < Dp * 16; e += (64 * NW)) {
            const int r = e >> 4, pp = e & 15;
            float v = 0.0f;
            if (r < D && row0 + pp < n) {
                const int64_t row = idx ? idx[row0 + pp] : row0 + pp;
                v = x[row * D + feat_of_rank[r]];
            }
            Xc[lidx(r, pp)] = v;
            xt[lidx(r, pp)] = v;
        }
        PHASE_END(1)

        // --------------------------------------------------------- forward
        float ladj = 0.0f;
        for (int t = 0; t < T; ++t) {
            const MafView wvw = maf_view(m, t);
            float* xtn = xt + (size_t)(t + 1) * Dp * 16;
            hidden_pass_wg<NW, PF, PROF>(m, wvw, Xc, A, B, Cb, wv, lane, pacc, tk);
            if (act) {                                       // A, B, Cb are contiguous in LDS: one copy
                float4* dst = reinterpret_cast<float4*>(act + (size_t)t * 3 * Hp * 16);
                for (int e = tid; e < 3 * Hp * 4; e += (64 * NW)) dst[e] = reinterpret_cast<const float4*>(A)[e];
            }
            LAPT(2)
            if (UNI == 0) {
                for (int O = wv; O < nOeff; O += NW) {
                    f32x4 o = bias4(wvw.b3, 16 * O + 4 * q);
                    o = mac_range<PF>(o, wvw.f3 + (size_t)O * nT * 64, Cb, 0, nT, lane);
                    for (int s = 0; s < 2; ++s) {
                        const int rank = 8 * O + 2 * q + s;
                        if (rank < D) {
                            const float shift = s ? o[2] : o[0];
                            const float ls = soft_ls(s ? o[3] : o[1]);
                            const float y = Xc[lidx(rank, p)] * expf(ls) + shift;
                            // the next transform reads its input by its own rank order
                            const int r2 = (t + 1 < T) ? rank_of_feat[(t + 1) * D + feat_of_rank[t * D + rank]] : rank;
                            Xn[lidx(r2, p)] = y;
                            xtn[lidx(r2, p)] = y;
                            ladj += ls;
                        }
                    }
                }
            } else {
                for (int c = 0; c < nXT; ++c) {
                    rqs_panel_train<NW, PF>(m, wvw, Cb, P, c, wv, lane,
                                    par ? par + ((size_t)t * nXT + c) * RQS_NOUT * 256 : nullptr);
                    lds_barrier();
                    for (int e = tid; e < 256; e += (64 * NW)) {
                        const int rr = e >> 4, pp = e & 15, rank = 16 * c + rr;
                        if (rank < D) {
                            float phi[RQS_NOUT];
#pragma unroll
                            for (int j = 0; j < RQS_NOUT; ++j) phi[j] = P[lidx(RQS_NOUT * rr + j, pp)];
                            float y, l;
                            rqs_forward(phi, Xc[lidx(rank, pp)], y, l);
                            const int r2 = (t + 1 < T) ? rank_of_feat[(t + 1) * D + feat_of_rank[t * D + rank]] : rank;
                            Xn[lidx(r2, pp)] = y;
                            xtn[lidx(r2, pp)] = y;
                            ladj += l;
                        }
                    }
                    lds_barrier();
                }
            }
            for (int e = tid; e < (Dp - D) * 16; e += (64 * NW)) {
                Xn[lidx(D + (e >> 4), e & 15)] = 0.0f;
                xtn[lidx(D + (e >> 4), e & 15)] = 0.0f;
            }
            PHASE_END(3)
            float* sw = Xc; Xc = Xn; Xn = sw;
        }
        // ------------------------------------------------------------ loss
        {
            const float l = quad_sum(ladj);
            if (lane < 16) RED[wv * 16 + lane] = l;
            __syncthreads();                             // full barrier: the xt scratch written above is read below
            const float* Z = Xc;                         // rank order of the last transform
            if (wv == 0) {
                float ss = 0.0f;
                for (int r = q; r < D; r += 4) { const float z = Z[lidx(r, p)]; ss += z * z; }
                ss = quad_sum(ss);
                float lt = 0.0f;
                for (int k = 0; k < NW; ++k) lt += RED[16 * k + p];
                float term = 0.0f;
                if (row0 + p < n) term = -CC[p] * ((-0.5f * ss - 0.9189385332046727f * (float)D) + lt);
                term += __shfl_xor(term, 1); term += __shfl_xor(term, 2);
                term += __shfl_xor(term, 4); term += __shfl_xor(term, 8);
                loss_acc += term;
            }
            __syncthreads();                             // Z may be Gb itself: finish reading it first
            // dL/dz = c * z
            for (int e = tid; e < Dp * 16; e += (64 * NW)) {
                const int r = e >> 4, pp = e & 15;
                Gb[lidx(r, pp)] = (r < D) ? CC[pp] * Z[lidx(r, pp)] : 0.0f;
            }
        }
        PHASE_END(1)

        // -------------------------------------------------------- backward
        for (int t = T - 1; t >= 0; --t) {
            const MafView wvw = maf_view(m, t);
            const TrainView tv = train_view(m, tr, t);
            const float4* xsrc = reinterpret_cast<const float4*>(xt + (size_t)t * Dp * 16);
            if (UNI == 0) {
                for (int e = tid; e < Dp * 4; e += (64 * NW)) reinterpret_cast<float4*>(E)[e] = xsrc[e];
                if (act) {
                    const float4* src = reinterpret_cast<const float4*>(act + (size_t)t * 3 * Hp * 16);
                    for (int e = tid; e < 3 * Hp * 4; e += (64 * NW)) reinterpret_cast<float4*>(A)[e] = src[e];
                    lds_barrier();
                } else {
                    lds_barrier();
                    // recompute this transform's activations
                    hidden_pass_wg<NW, PF, PROF>(m, wvw, E, A, B, Cb, wv, lane, pacc, tk);
                }
                // (shift, raw) of the transform
                for (int O = wv; O < nOeff; O += NW) {
                    f32x4 o = bias4(wvw.b3, 16 * O + 4 * q);
                    o = mac_range<PF>(o, wvw.f3 + (size_t)O * nT * 64, Cb, 0, nT, lane);
                    store_rows(P, O, q, p, o);
                }
                PHASE_END(4)
                // element-wise part: y = x e^{ls} + shift,  L += -c * sum ls   (in place: P -> dP, G -> direct dx)
                for (int e = tid; e < Dp * 16; e += (64 * NW)) {
                    const int r = e >> 4, pp = e & 15;
                    float gs = 0.0f, gr = 0.0f, gx = 0.0f;
                    if (r < D) {
                        const float xv = E[lidx(r, pp)];
                        const float raw = P[lidx(2 * r + 1, pp)];
                        const float den = 1.0f + fabsf(raw / PMC_LOG_SLOPE);
                        const float el = expf(raw / den);
                        const float gy = Gb[lidx(r, pp)];
                        gs = gy;
                        gr = (gy * xv * el - CC[pp]) / (den * den);
                        gx = gy * el;
                    }
                    P[lidx(2 * r, pp)] = gs;
                    P[lidx(2 * r + 1, pp)] = gr;
                    Gb[lidx(r, pp)] = gx;
                }
                PHASE_END(5)
                // ---- layer 3: da2 = relu'(h2) . W3^T dP -> E ; dW3, db3, db2
                for (int K = wv; K < nT; K += NW) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f};
                    a = mac_range<PF>(a, tv.f3T + (size_t)K * nOT * 64, P, 0, nOeff, lane);
                    a = relu_gate(a, Cb, K, q, p);
                    store_rows(E, K, q, p, a);
                    bias_put(slab + tv.gb2 + 16 * K + 4 * q, a, lane, first);
                }
                for (int O = wv; O < nOeff; O += NW)
                    bias_put(slab + tv.gb3 + 16 * O + 4 * q, rows_of(P, O, q, p), lane, first);
                for (int i = dw3_start, O = 0, K = dw3_start; i < dw3_start + dw3_count; ++i, ++K) {
                    while (K >= nT) { K -= nT; ++O; }              // (O, K) of tile i without a division
                    slab_put4(slab + tv.g3 + ((size_t)i * 64 + lane) * 4, outer_tile(P, O, Cb, K, lane), first);
                }
                PHASE_END(6)
            } else {
                // x_t -> XB (kept for the whole spline sweep), E <- 0 (accumulates W3^T dP over the panels)
                for (int e = tid; e < Dp * 4; e += (64 * NW)) reinterpret_cast<float4*>(XB)[e] = xsrc[e];
                for (int e = tid; e < Hp * 4; e += (64 * NW))
                    reinterpret_cast<float4*>(E)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (act) {
                    const float4* src = reinterpret_cast<const float4*>(act + (size_t)t * 3 * Hp * 16);
                    for (int e = tid; e < 3 * Hp * 4; e += (64 * NW)) reinterpret_cast<float4*>(A)[e] = src[e];
                    lds_barrier();
                } else {
                    lds_barrier();
                    hidden_pass_wg<NW, PF, PROF>(m, wvw, XB, A, B, Cb, wv, lane, pacc, tk);
                }
                LAPT(4)
                for (int c = 0; c < nXT; ++c) {
                    const int O0 = RQS_NOUT * c;                         // first output tile of the panel
                    const int nO = min(RQS_NOUT, nOeff - O0);            // its tiles with real rows
                    rqs_panel_train<NW, PF>(m, wvw, Cb, P, c, wv, lane, nullptr,
                                    par ? par + ((size_t)t * nXT + c) * RQS_NOUT * 256 : nullptr);
                    PHASE_END(4)
                    // spline backward in place: P -> dP, G -> direct dL/dx term
                    for (int e = tid; e < 256; e += (64 * NW)) {
                        const int rr = e >> 4, pp = e & 15, rank = 16 * c + rr;
                        if (rank < D) {
                            float phi[RQS_NOUT], dphi[RQS_NOUT];
#pragma unroll
                            for (int j = 0; j < RQS_NOUT; ++j) phi[j] = P[lidx(RQS_NOUT * rr + j, pp)];
                            float gx;
                            rqs_backward(phi, XB[lidx(rank, pp)], Gb[lidx(rank, pp)], -CC[pp], dphi, gx);
#pragma unroll
                            for (int j = 0; j < RQS_NOUT; ++j) P[lidx(RQS_NOUT * rr + j, pp)] = dphi[j];
                            Gb[lidx(rank, pp)] = gx;
                        }
                    }
                    PHASE_END(5)
                    // W3^T dP of this panel into E (every wave owns whole tiles of E), db3, dW3
                    for (int K = wv; K < nT; K += NW) {
                        f32x4 a = rows_of(E, K, q, p);
                        a = mac_range<PF>(a, tv.f3T + ((size_t)K * nOT + O0) * 64, P, 0, nO, lane);
                        store_rows(E, K, q, p, a);
                    }
                    for (int i = wv; i < nO; i += NW)
                        bias_put(slab + tv.gb3 + 16 * (O0 + i) + 4 * q, rows_of(P, i, q, p), lane, first);
                    {
                        // dW3 tiles (Ol, K) of the panel.  The waves wv < nT just multiplied a whole tile of E (nO K
                        // steps each): with fewer hidden tiles than waves they take only qh weight-gradient tiles
                        // (a tile costs about two K steps), the free waves share the rest.
                        const int n_light = nO * nT;
                        int first_i = wv, stride = NW, last_i = n_light;
                        if (nT < NW) {
                            const int qh = max(0, ((nT * nO + 2 * n_light) / NW - nO) / 2);
                            if (wv < nT) { stride = nT; last_i = min(n_light, nT * qh); }
                            else { first_i = nT * qh + (wv - nT); stride = NW - nT; }
                        }
                        for (int i = first_i; i < last_i; i += stride) {
                            const int Ol = i / nT, K = i - Ol * nT;
                            slab_put4(slab + tv.g3 + (((size_t)(O0 + Ol) * nT + K) * 64 + lane) * 4,
                                      outer_tile(P, Ol, Cb, K, lane), first);
                        }
                    }
                    PHASE_END(6)
                }
                // da2 = relu'(h2) . (W3^T dP), db2
                for (int K = wv; K < nT; K += NW) {
                    f32x4 a = relu_gate(rows_of(E, K, q, p), Cb, K, q, p);
                    store_rows(E, K, q, p, a);
                    bias_put(slab + tv.gb2 + 16 * K + 4 * q, a, lane, first);
                }
                PHASE_END(6)
            }
            // ---- layer 2: da1 = relu'(h1) . (da2 + W2^T da2) -> C ; dW2 (da2 x h1), db1
            for (int it = 0;; ++it) {                              // the tiles this wave owns, most expensive first
                const int Ti = snake_item<NW>(wv, it);
                if (Ti >= nT) break;
                f32x4 a = rows_of(E, Ti, q, p);
                a = mac_range<PF>(a, tv.f2T + (size_t)Ti * nT * 64, E, (m.tri_ok ? Ti : 0), nT, lane);
                a = relu_gate(a, B, Ti, q, p);
                store_rows(Cb, Ti, q, p, a);
                bias_put(slab + tv.gb1 + 16 * Ti + 4 * q, a, lane, first);
            }
            {   // weight-gradient tiles (To, Ti <= To) in row-major order: this wave's contiguous range
                int To = 0, Ti = dwt_start;
                for (int i = 0; i < dwt_count; ++i, ++Ti) {
                    if (m.tri_ok) { while (Ti > To) { Ti -= To + 1; ++To; } }
                    else { while (Ti >= nT) { Ti -= nT; ++To; } }
                    slab_put4(slab + tv.g2 + (((size_t)To * nT + Ti) * 64 + lane) * 4, outer_tile(E, To, B, Ti, lane),
                              first);
                }
            }
            PHASE_END(7)
            // ---- layer 1: da0 = relu'(h0) . (da1 + W1^T da1) -> E ; dW1 (da1 x h0), db0 ; x_t -> B
            for (int it = 0;; ++it) {
                const int Ti = snake_item<NW>(wv, it);
                if (Ti >= nT) break;
                f32x4 a = rows_of(Cb, Ti, q, p);
                a = mac_range<PF>(a, tv.f1T + (size_t)Ti * nT * 64, Cb, (m.tri_ok ? Ti : 0), nT, lane);
                a = relu_gate(a, A, Ti, q, p);
                store_rows(E, Ti, q, p, a);
                bias_put(slab + tv.gb0 + 16 * Ti + 4 * q, a, lane, first);
            }
            for (int e = tid; e < Dp * 4; e += (64 * NW)) reinterpret_cast<float4*>(B)[e] = xsrc[e];
            {   // weight-gradient tiles (To, Ti <= To) in row-major order: this wave's contiguous range
                int To = 0, Ti = dwt_start;
                for (int i = 0; i < dwt_count; ++i, ++Ti) {
                    if (m.tri_ok) { while (Ti > To) { Ti -= To + 1; ++To; } }
                    else { while (Ti >= nT) { Ti -= nT; ++To; } }
                    slab_put4(slab + tv.g1 + (((size_t)To * nT + Ti) * 64 + lane) * 4, outer_tile(Cb, To, A, Ti, lane),
                              first);
                }
            }
            PHASE_END(8)
            // ---- layer 0: dW0 (da0 x x), dx = direct + W0^T da0 -> P
            if (t > 0) {
                for (int Xi = wv; Xi < nXT; Xi += NW) {
                    f32x4 a = rows_of(Gb, Xi, q, p);
                    a = mac_range<PF>(a, tv.f0T + (size_t)Xi * nT * 64, E, 0, nT, lane);
                    store_rows(P, Xi, q, p, a);
                }
            }
            {
                // without the dx tiles (t == 0) the ranges are still a valid partition, just less balanced
                for (int i = dw0_start, To = 0, Xi = dw0_start; i < dw0_start + dw0_count; ++i, ++Xi) {
                    while (Xi >= nXT) { Xi -= nXT; ++To; }
                    slab_put4(slab + tv.g0 + ((size_t)i * 64 + lane) * 4, outer_tile(E, To, B, Xi, lane), first);
                }
            }
            PHASE_END(9)
            if (t > 0) {
                // re-rank for transform t-1 (its output order)
                for (int e = tid; e < Dp * 16; e += (64 * NW)) {
                    const int r = e >> 4, pp = e & 15;
                    if (r < D) Gb[lidx(rank_of_feat[(t - 1) * D + feat_of_rank[t * D + r]], pp)] = P[lidx(r, pp)];
                }
                PHASE_END(9)
            }
        }
    }
    if (TRAIN_WARM_L2) asm volatile("" :: "v"(warm));
    if (tid == 0) tr.loss_partial[blockIdx.x] = loss_acc;
    if (PROF && lane == 0) {
        pacc[0] = TICKT() - t_begin;
        long long* o = prof + ((size_t)blockIdx.x * NW + wv) * 16;
        for (int i = 0; i < 16; ++i) o[i] = pacc[i];
    }
}

// grad[gmap[i]] = sum over slabs of slab[i]; per-block sum of squares; loss += sum of the per-workgroup losses
__global__ __launch_bounds__(256) void reduce_slabs_kernel(pmc_maf_train_t tr, int n_slabs, int64_t g_total,
                                                           float* __restrict__ grad, float* __restrict__ loss) {
    __shared__ float red[4];
    const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float sq = 0.0f;
    if (i4 * 4 < g_total) {
        const int4 g = reinterpret_cast<const int4*>(tr.gmap)[i4];
        if ((g.x & g.y & g.z & g.w) >= 0) {               // any element mapped
            float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
            const float4* src = reinterpret_cast<const float4*>(tr.slabs) + i4;
            const size_t stride4 = (size_t)tr.slab_stride / 4;
            int b = 0;
            for (; b + 1 < n_slabs; b += 2) {
                const float4 u = src[(size_t)b * stride4], v = src[(size_t)(b + 1) * stride4];
                s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
                s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
            }
            if (b < n_slabs) {
                const float4 u = src[(size_t)b * stride4];
                s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
            }
            s0.x += s1.x; s0.y += s1.y; s0.z += s1.z; s0.w += s1.w;
            if (g.x >= 0) { grad[g.x] = s0.x; sq += s0.x * s0.x; }
            if (g.y >= 0) { grad[g.y] = s0.y; sq += s0.y * s0.y; }
            if (g.z >= 0) { grad[g.z] = s0.z; sq += s0.z * s0.z; }
            if (g.w >= 0) { grad[g.w] = s0.w; sq += s0.w * s0.w; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        tr.sq_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        if (blockIdx.x == 0 && loss) {
            float s = 0.0f;
            for (int b = 0; b < n_slabs; ++b) s += tr.loss_partial[b];
            *loss += s;
        }
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
    if (threadIdx.x == 0) *out += (red[0] + red[1]) + (red[2] + red[3]);      // ONE block: a fixed order of additions
}

// validation loss of one batch (flow.py:336-341): loss += sum_i -(logp_i * c_i), c_i = 1 or
// w[idx_i] * wmul / sum_j w[idx_j]; ONE block, fixed order (deterministic)
__global__ __launch_bounds__(256) void batch_nll_kernel(const float* __restrict__ logp, const float* __restrict__ w,
                                                        const int64_t* __restrict__ idx, float wmul,
                                                        float* __restrict__ loss, int64_t n) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    float scale = 1.0f;
    if (w) {
        float s = 0.0f;
        for (int64_t i = tid; i < n; i += 256) s += w[idx ? idx[i] : i];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        scale = wmul / ((red[0] + red[1]) + (red[2] + red[3]));
        __syncthreads();
    }
    float s = 0.0f;
    for (int64_t i = tid; i < n; i += 256) s += -(logp[i] * (w ? w[idx ? idx[i] : i] * scale : 1.0f));
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) *loss += (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sum_kernel(const float* __restrict__ v, float* __restrict__ out, int64_t n) {
    __shared__ float red[4];
    float s = 0.0f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) s += v[e];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out += (red[0] + red[1]) + (red[2] + red[3]);      // ONE block: a fixed order of additions
}

// ---------------------------------------------------------------------------
// clip_grad_norm_ (flow.py:318) + AdamW (flow.py:268, :319)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, float* __restrict__ part,
                                                             int64_t n) {
    __shared__ float red[4];
    float s = 0.0f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) s += g[e] * g[e];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ mo, float* __restrict__ vo, int64_t n,
                                                    float lr, float b1, float b2, float eps, float wd, float max_norm,
                                                    float bc1, float bc2, const float* __restrict__ sq_part, int n_part,
                                                    const int* __restrict__ sc_ptr, const int* __restrict__ sc_dst,
                                                    float* __restrict__ img_a, int n_a, float* __restrict__ img_b) {
    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
    __shared__ float red[4];
    float coef = 1.0f;
    if (max_norm > 0.0f) {
        float s = 0.0f;                                   // every block adds the partials in the same order
        for (int i = threadIdx.x; i < n_part; i += 256) s += sq_part[i];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        coef = max_norm / (sqrtf((red[0] + red[1]) + (red[2] + red[3])) + 1e-6f);
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
        if (sc_ptr) {
            // refresh of the kernel images (pmc_adamw_t.scatter_*): every place this parameter is packed to
            for (int k = sc_ptr[e], k1 = sc_ptr[e + 1]; k < k1; ++k) {
                const int d = sc_dst[k];
                if (d < n_a) img_a[d] = pv; else img_b[d - n_a] = pv;
            }
        }
    }
}

// both kernel images from the canonical vector in one launch
__global__ __launch_bounds__(256) void pack2_kernel(const float* __restrict__ flat, const int* __restrict__ idx_a,
                                                    float* __restrict__ dst_a, int64_t n_a,
                                                    const int* __restrict__ idx_b, float* __restrict__ dst_b,
                                                    int64_t n_b) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_a + n_b; e += (int64_t)gridDim.x * 256) {
        if (e < n_a) { const int i = idx_a[e]; dst_a[e] = i >= 0 ? flat[i] : 0.0f; }
        else { const int64_t f = e - n_a; const int i = idx_b[f]; dst_b[f] = i >= 0 ? flat[i] : 0.0f; }
    }
}

// ---------------------------------------------------------------------------
static size_t train_lds_bytes(const pmc_maf_t& m) {
    if (m.n_out == RQS_NOUT)
        return (size_t)(4 * m.Hp * 16 + RQS_NOUT * 256 + 2 * m.Dp * 16 + 16 + 16 * TRAIN_WAVES) * sizeof(float);
    return (size_t)(4 * m.Hp * 16 + 2 * m.Dp * 16 + m.Dp * 16 + 16 + 16 * TRAIN_WAVES) * sizeof(float);   // (sized for 16 waves)
}

static int train_check(const pmc_maf_t* m, const pmc_maf_train_t* tr, const char* who) {
    if (!m || !tr || !tr->packedT || !tr->gmap || !tr->slabs || !tr->xt_scratch || !tr->loss_partial ||
        !tr->sq_partial || !tr->sched || tr->sched_waves != train_waves_of(*m) || tr->n_slabs < 1 || tr->slab_stride < (int64_t)m->T * tr->gmap_per_transform ||
        (tr->slab_stride & 3))
        return pmc_fail((std::string(who) + ": incomplete training image").c_str());
    return 0;
}

static int launch_lossgrad(const pmc_maf_t* m, const pmc_maf_train_t* tr, const float* x, const float* w,
                           const int64_t* idx, float wmul, float* grad, float* loss, int64_t n, hipStream_t st,
                           long long* prof = nullptr) {
    const size_t lds = train_lds_bytes(*m);
    if (lds > 160 * 1024) return pmc_fail("pmc_maf_loss_grad: flow too wide for the 160 KB LDS of one workgroup");
    static size_t lds_set = 0;
    if (lds > 48 * 1024 && lds > lds_set) {
        hipError_t e = hipSuccess;
        const void* ks[6] = {reinterpret_cast<const void*>(maf_lossgrad_kernel<TRAIN_WAVES, TRAIN_PF, false, 0>),
                             reinterpret_cast<const void*>(maf_lossgrad_kernel<TRAIN_WAVES, TRAIN_PF, true, 0>),
                             reinterpret_cast<const void*>(maf_lossgrad_kernel<TRAIN_WAVES, TRAIN_PF, false, 1>),
                             reinterpret_cast<const void*>(maf_lossgrad_kernel<TRAIN_WAVES, TRAIN_PF, true, 1>),
                             reinterpret_cast<const void*>(maf_lossgrad_kernel<TRAIN_WAVES_NARROW, TRAIN_PF_NARROW, false, 1>),
                             reinterpret_cast<const void*>(maf_lossgrad_kernel<TRAIN_WAVES_NARROW, TRAIN_PF_NARROW, true, 1>)};
        for (int i = 0; i < 6 && e == hipSuccess; ++i)
            e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pmc_fail_hip(e, "hipFuncSetAttribute(maf_lossgrad_kernel)");
        lds_set = lds;
    }
    const int64_t nsets = (n + 15) / 16;
    const int n_wg = (int)(nsets < tr->n_slabs ? nsets : tr->n_slabs);
    const bool rqs = (m->n_out == RQS_NOUT);
    const bool narrow = train_waves_of(*m) == TRAIN_WAVES_NARROW;
#define LG(NWV, PFV, PR, UN)                                                                                       \
    hipLaunchKernelGGL((maf_lossgrad_kernel<NWV, PFV, PR, UN>), dim3((unsigned)n_wg), dim3(64 * NWV), lds, st, *m, *tr, x, \
                       w, idx, wmul, n, prof)
    if (narrow) { if (prof) LG(TRAIN_WAVES_NARROW, TRAIN_PF_NARROW, true, 1); else LG(TRAIN_WAVES_NARROW, TRAIN_PF_NARROW, false, 1); }
    else if (rqs) { if (prof) LG(TRAIN_WAVES, TRAIN_PF, true, 1); else LG(TRAIN_WAVES, TRAIN_PF, false, 1); }
    else { if (prof) LG(TRAIN_WAVES, TRAIN_PF, true, 0); else LG(TRAIN_WAVES, TRAIN_PF, false, 0); }
#undef LG
    const int64_t g_total = (int64_t)m->T * tr->gmap_per_transform;
    const int64_t blocks = (g_total / 4 + 255) / 256;
    if (blocks > tr->n_sq_partial) return pmc_fail("pmc_maf_loss_grad: sq_partial too small");
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, st, *tr, n_wg, g_total, grad, loss);
    return pmc_check_launch("maf_lossgrad_kernel");
}

extern "C" int pmc_maf_loss_grad(const pmc_maf_t* m, const pmc_maf_train_t* tr, const float* x, const float* w,
                                 const int64_t* idx, float wmul, float* grad, float* loss, int64_t n, void* stream) {
    if (train_check(m, tr, "pmc_maf_loss_grad")) return 1;
    if (!x || !grad || !loss || n < 0) return pmc_fail("pmc_maf_loss_grad: bad argument");
    if (n == 0) return 0;
    return launch_lossgrad(m, tr, x, w, idx, wmul, grad, loss, n, (hipStream_t)stream);
}

// in-kernel cycle profile (scripts/profile_train.py; not part of the ABI): prof i64 [n_wg][TRAIN_WAVES][16]
extern "C" int pmc_debug_lossgrad_profile(const pmc_maf_t* m, const pmc_maf_train_t* tr, const float* x, float* grad,
                                          float* loss, int64_t n, long long* prof, void* stream) {
    if (train_check(m, tr, "pmc_debug_lossgrad_profile")) return 1;
    return launch_lossgrad(m, tr, x, nullptr, nullptr, 1000.0f, grad, loss, n, (hipStream_t)stream, prof);
}
// waves per training workgroup for this flow = the n_waves of pmc_maf_train_t.sched (MAFSpec.train_schedule)
extern "C" int pmc_maf_train_waves(const pmc_maf_t* m) { return m ? train_waves_of(*m) : TRAIN_WAVES; }

// One pass over a validation set in batches (flow.py:327-348), everything enqueued by one call.
extern "C" int pmc_maf_valid_epoch(const pmc_maf_t* m, const float* x, const float* w, const int64_t* perm, int64_t n,
                                   int64_t batch_size, float* logp_scratch, float* loss, void* stream) {
    if (!m || !x || !logp_scratch || !loss || n < 0 || batch_size < 1) return pmc_fail("pmc_maf_valid_epoch: bad argument");
    hipStream_t st = (hipStream_t)stream;
    for (int64_t b0 = 0; b0 < n; b0 += batch_size) {
        const int64_t nb = (n - b0 < batch_size) ? n - b0 : batch_size;
        const float* xb = perm ? x : x + b0 * m->D;
        const float* wb = (w && !perm) ? w + b0 : w;
        const int64_t* ib = perm ? perm + b0 : nullptr;
        if (int rc = pmc_launch_forward_wg(m, xb, nullptr, nullptr, logp_scratch, nb, st, ib)) return rc;
        hipLaunchKernelGGL(batch_nll_kernel, dim3(1), dim3(256), 0, st, (const float*)logp_scratch, wb, ib, 1000.0f, loss, nb);
    }
    return pmc_check_launch("pmc_maf_valid_epoch");
}

extern "C" int pmc_neg_weighted_sum(const float* logp, const float* w, const float* wsum, float wmul, float* out,
                                    int64_t n, void* stream) {
    if (!logp || !out || n < 0 || (w && !wsum)) return pmc_fail("pmc_neg_weighted_sum: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(neg_weighted_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logp, w, wsum,
                       wmul, out, n);
    return pmc_check_launch("neg_weighted_sum_kernel");
}

extern "C" int pmc_sum_f32(const float* v, float* out, int64_t n, void* stream) {
    if (!v || !out || n < 0) return pmc_fail("pmc_sum_f32: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, v, out, n);
    return pmc_check_launch("sum_kernel");
}

static void launch_adamw(float* params, const float* grad, float* m1, float* m2, int64_t n, double lr, double beta1,
                         double beta2, double eps, double wd, double max_norm, int64_t step, const float* sq_part,
                         int n_part, hipStream_t st, const int* sc_ptr = nullptr, const int* sc_dst = nullptr,
                         float* img_a = nullptr, int n_a = 0, float* img_b = nullptr) {
    int64_t grid = (n + 255) / 256; if (grid > 512) grid = 512;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)grid), dim3(256), 0, st, params, grad, m1, m2, n, (float)lr,
                       (float)beta1, (float)beta2, (float)eps, (float)wd, (float)max_norm, (float)bc1, (float)bc2,
                       sq_part, n_part, sc_ptr, sc_dst, img_a, n_a, img_b);
}

// (for maf_train_bf16.hip) sum-of-squares partials + clipped AdamW step without image refresh
int pmc_launch_clip_adamw(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                          double beta1, double beta2, double eps, double wd, double max_norm, int64_t step,
                          float* sq_scratch, hipStream_t st) {
    const int n_part = PMC_ADAMW_SCRATCH;
    if (max_norm > 0.0)
        hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(n_part), dim3(256), 0, st, grad, sq_scratch, n);
    launch_adamw(params, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, wd, max_norm, step, sq_scratch, n_part, st);
    return pmc_check_launch("adamw_kernel");
}

extern "C" int pmc_adamw_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                              double lr, double beta1, double beta2, double eps, double weight_decay,
                              double max_norm, int64_t step, float* sq_scratch, void* stream) {
    if (!params || !grad || !exp_avg || !exp_avg_sq || !sq_scratch || n <= 0 || step < 1)
        return pmc_fail("pmc_adamw_step: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int n_part = PMC_ADAMW_SCRATCH;
    if (max_norm > 0.0)
        hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(n_part), dim3(256), 0, st, grad, sq_scratch, n);
    launch_adamw(params, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, max_norm, step,
                 sq_scratch, n_part, st);
    return pmc_check_launch("adamw_kernel");
}

extern "C" int pmc_maf_train_epoch(const pmc_maf_t* m, const pmc_maf_train_t* tr, pmc_adamw_t* opt, const float* x,
                                   const float* w, const int64_t* perm, int64_t n, int64_t batch_size, float* loss,
                                   void* stream) {
    if (train_check(m, tr, "pmc_maf_train_epoch")) return 1;
    if (!opt || !opt->params || !opt->grad || !opt->exp_avg || !opt->exp_avg_sq || !opt->pack_idx || !opt->packed ||
        !opt->packT_idx || !opt->packedT || opt->n_params <= 0 || !x || !loss || n < 0 || batch_size < 1)
        return pmc_fail("pmc_maf_train_epoch: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t g_total = (int64_t)m->T * tr->gmap_per_transform;
    const int n_part = (int)((g_total / 4 + 255) / 256);
    for (int64_t b0 = 0; b0 < n; b0 += batch_size) {
        const int64_t nb = (n - b0 < batch_size) ? n - b0 : batch_size;
        // a batch is rows perm[b0 .. b0+nb) of x, or rows b0 .. b0+nb when perm == NULL
        const float* xb = perm ? x : x + b0 * m->D;
        const float* wb = (w && !perm) ? w + b0 : w;
        if (launch_lossgrad(m, tr, xb, wb, perm ? perm + b0 : nullptr, 1000.0f, opt->grad, loss, nb, st)) return 1;
        opt->step += 1;
        const bool scatter = opt->scatter_ptr && opt->scatter_dst && opt->n_packed < 0x7fffffffLL;
        launch_adamw(opt->params, opt->grad, opt->exp_avg, opt->exp_avg_sq, opt->n_params, opt->lr, opt->beta1,
                     opt->beta2, opt->eps, opt->weight_decay, opt->max_norm, opt->step, tr->sq_partial, n_part, st,
                     scatter ? opt->scatter_ptr : nullptr, opt->scatter_dst, opt->packed, (int)opt->n_packed,
                     opt->packedT);
        if (!scatter) {
            const int64_t tot = opt->n_packed + opt->n_packedT;
            int64_t grid = (tot + 255) / 256; if (grid > 2048) grid = 2048;
            hipLaunchKernelGGL(pack2_kernel, dim3((unsigned)grid), dim3(256), 0, st, opt->params, opt->pack_idx,
                               opt->packed, opt->n_packed, opt->packT_idx, opt->packedT, opt->n_packedT);
        }
    }
    return pmc_check_launch("pmc_maf_train_epoch");
}

// ---------------------------------------------------------------------------
// Flow.fit options: weight regularisation (flow.py:314-315, :387-421) and noise augmentation (flow.py:240-245, :304-307)
// ---------------------------------------------------------------------------
#include "philox.h"

// R = sum_{weight entries} |p| / laplace + p^2 / (2 gaussian^2); grad (optional) += dR/dp.  256 blocks, contiguous
// chunks, partials summed in block order by penalty_final_kernel.
__global__ __launch_bounds__(256) void penalty_kernel(const float* __restrict__ p, const uint8_t* __restrict__ is_w,
                                                      float* __restrict__ grad, int64_t n, float inv_b, float inv_s2,
                                                      float* __restrict__ part) {
    __shared__ float red[4];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    float r = 0.0f;
    for (int64_t e = lo + threadIdx.x; e < hi; e += 256) {
        if (is_w[e]) {
            const float v = p[e];
            r += fabsf(v) * inv_b + 0.5f * v * v * inv_s2;
            if (grad) grad[e] += (v > 0.0f ? inv_b : (v < 0.0f ? -inv_b : 0.0f)) + v * inv_s2;
        }
    }
    for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = r;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void penalty_final_kernel(const float* __restrict__ part, int n_part, float mult, float* __restrict__ loss) {
    if (threadIdx.x == 0) {
        float s = 0.0f;
        for (int i = 0; i < n_part; ++i) s += part[i];
        *loss += mult * s;
    }
}

extern "C" int pmc_weight_penalty(const float* params, const uint8_t* is_weight, float* grad, int64_t n,
                                  double laplace_scale, double gaussian_scale, float mult, float* loss,
                                  float* scratch, void* stream) {
    if (!params || !is_weight || !loss || !scratch || n < 1) return pmc_fail("pmc_weight_penalty: bad argument");
    const float inv_b = laplace_scale > 0.0 ? (float)(1.0 / laplace_scale) : 0.0f;
    const float inv_s2 = gaussian_scale > 0.0 ? (float)(1.0 / (gaussian_scale * gaussian_scale)) : 0.0f;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(penalty_kernel, dim3(PMC_ADAMW_SCRATCH), dim3(256), 0, st, params, is_weight, grad, n, inv_b, inv_s2, scratch);
    hipLaunchKernelGGL(penalty_final_kernel, dim3(1), dim3(64), 0, st, (const float*)scratch, (int)PMC_ADAMW_SCRATCH, mult, loss);
    return pmc_check_launch("penalty_kernel");
}

// out = x + scale * N(0, 1): torch.randn_like noise of flow.py:305 / :334, Philox keyed by (seed, pass, row, pair)
__global__ __launch_bounds__(256) void add_noise_kernel(const float* __restrict__ x, int64_t n, int D, float scale,
                                                        uint64_t seed, uint64_t pass, uint64_t row0, float* __restrict__ out) {
    const int half = (D + 1) / 2;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n * half; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / half; const int j = (int)(e % half) * 2;
        Philox ph(seed, pass, row0 + (uint64_t)r, 4);      // keyed by the GLOBAL row: a rank's shard draws what the whole set would
        ph.ctr[0] = (uint32_t)(j >> 1);
        double a, b;
        ph.normal2(a, b);
        out[r * D + j] = x[r * D + j] + scale * (float)a;
        if (j + 1 < D) out[r * D + j + 1] = x[r * D + j + 1] + scale * (float)b;
    }
}

extern "C" int pmc_add_noise_rows_f32(const float* x, int64_t n, int32_t D, float scale, uint64_t seed, uint64_t pass,
                                      uint64_t row0, float* out, void* stream) {
    if (!x || !out || n < 1 || D < 1) return pmc_fail("pmc_add_noise_f32: bad argument");
    int64_t grid = (n * ((D + 1) / 2) + 255) / 256; if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(add_noise_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, n, (int)D, scale, seed, pass,
                       row0, out);
    return pmc_check_launch("add_noise_kernel");
}

extern "C" int pmc_add_noise_f32(const float* x, int64_t n, int32_t D, float scale, uint64_t seed, uint64_t pass,
                                 float* out, void* stream) {
    return pmc_add_noise_rows_f32(x, n, D, scale, seed, pass, 0, out, stream);
}

// out[0] = mean_j || x[row] - x[j] ||_2   (flow.py:241-245: the quantity the reference's noise scale is built from)
__global__ __launch_bounds__(1024) void mean_distance_kernel(const float* __restrict__ x, int64_t n, int D, int64_t row,
                                                             float* __restrict__ out) {
    __shared__ float part[1024];
    const int64_t per = (n + 1023) / 1024;
    const int64_t lo = (int64_t)threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    float s = 0.0f;
    for (int64_t j = lo; j < hi; ++j) {
        float d2 = 0.0f;
        for (int k = 0; k < D; ++k) { const float d = x[row * D + k] - x[j * D + k]; d2 += d * d; }
        s += sqrtf(d2);
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0] / (float)n;
}

extern "C" int pmc_mean_distance_f32(const float* x, int64_t n, int32_t D, int64_t row, float* out, void* stream) {
    if (!x || !out || n < 1 || D < 1 || row < 0 || row >= n) return pmc_fail("pmc_mean_distance_f32: bad argument");
    hipLaunchKernelGGL(mean_distance_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, (int)D, row, out);
    return pmc_check_launch("mean_distance_kernel");
}

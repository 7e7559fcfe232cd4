// Flow training kernels (pocomc/flow.py:165-384): loss + parameter gradient of one minibatch,
// global-norm clip + AdamW, and the per-epoch driver.
//
//   loss = sum_n c_n * (-log_prob(x_n)),   c_n = 1                         (flow.py:309)
//                                          c_n = w_n * 1000 / sum(w_batch) (flow.py:311-312)
//
// A minibatch is at most 512 rows (sampler.py:289): 32 row sets of 16.  The step is two launches.
//
// (1) maf_chain_kernel -- the part that IS a dependency chain.  One WORKGROUP of TRAIN_WAVES (16; 8 for narrow
// spline flows) wavefronts owns 16 rows: the tiles of every layer are dealt to the waves, the activations sit in
// workgroup LDS in the MFMA operand layout of the inference kernels, and a barrier separates dependent layers.
// Forward keeps, per transform, its input, the three hidden activations and the hyper-network's outputs in a global
// scratch (L2 / MALL resident); the backward sweep takes one transform at a time:
//     d(shift, raw) -> dh2 = W3^T . -> relu' -> dh1 = da2 + W2^T da2 -> ...
// Data-gradient products  dh = W^T da  are MFMA bursts over pre-transposed weight fragments (packedT).  Every delta
// (d outputs, da2, da1, da0) is stored NEXT TO the activation it pairs with -- the workgroup computes no weight
// gradient: on the f32 matrix pipe of ONE compute unit (32 cycles per v_mfma_f32_16x16x4_f32 and SIMD) the
// weight-gradient tiles were a third of the chain's MFMA time, and the 32 workgroups of a 512-row batch leave
// 224 of the 256 compute units idle.
//
// (2) maf_dw_kernel -- the part that is NOT a chain.  dW[out][in] = sum over ALL rows of delta[out][row] *
// act[in][row] is a K = 512 product tiled over the WEIGHT matrix: one workgroup per 16 x 16 tile with at least one
// unmasked entry (MAFSpec.train_jobs: a few hundred per flow, on every compute unit), its four wavefronts take a
// quarter of the row sets each (four v_mfma_f32_16x16x4_f32 per set, operands straight from the scratch in the LDS
// layout the chain kernel wrote), partial tiles are added in wave order, the bias gradient of the tile's 16 output
// units rides along with the first tile of a row (the sum of the A operand over the rows), and the tile goes to the
// canonical gradient through the host-built map (masked weights and padding map to -1 and are never touched).  No
// atomics, no per-workgroup slabs, no reduction pass: sums run in a fixed order, a training run is bitwise
// reproducible.  Its block sums of squares feed the clip of the optimizer launch.
//
// LDS buffers alias along the backward sweep (4 hidden-width buffers instead of 7):
//     E: x_t -> da2 -> da0      A: h0      B: h1      C: h2 -> da1
//     P: (shift, raw) -> their gradients -> dx      G: dL/dy -> direct dL/dx term -> dL/dy of t-1
// which keeps the 8-transform, H=512, D=128 flow (BASELINE config 5) inside 160 KB.

#include <string>
#include "maf_common.h"
#include "maf_wg.h"
#include "rqs.h"

#ifndef TRAIN_PF
#define TRAIN_PF 4                  // weight fragments in flight per wave
#endif
#ifndef TRAIN_WAVES
#define TRAIN_WAVES 16              // waves of a training workgroup ...
#endif
// ... except for narrow spline flows (hidden width <= 64): their layers have fewer tiles than that many waves, the
// spline evaluation runs on four waves either way, and a barrier over 8 waves is cheaper.  With half the waves a
// wave has 256 registers: 8 fragments in flight.
#define TRAIN_WAVES_NARROW 8
#define TRAIN_PF_NARROW 8
static int train_waves_of(const pmc_maf_t& m) {
    return (m.n_out != 2 && m.nT <= 6) ? TRAIN_WAVES_NARROW : TRAIN_WAVES;     // spline flows of hidden width <= 64
}

struct TrainView {
    const float4* f0T; const float4* f1T; const float4* f2T; const float4* f3T;
};

__device__ __forceinline__ TrainView train_view(const pmc_maf_t& m, const pmc_maf_train_t& tr, int t) {
    TrainView v;
    const size_t nT = m.nT, nXT = m.nXT;
    const float* p = tr.packedT + (size_t)t * tr.pkT_per_transform;
    v.f0T = reinterpret_cast<const float4*>(p); p += nXT * nT * 256;
    v.f1T = reinterpret_cast<const float4*>(p); p += nT * nT * 256;
    v.f2T = reinterpret_cast<const float4*>(p); p += nT * nT * 256;
    v.f3T = reinterpret_cast<const float4*>(p);
    return v;
}

#define TRAIN_OWN_MAX 8            // hidden tiles a wave can own through the table (more: the snake deal)

// hidden layers of one transform for the workgroup's 16 rows: X -> H0, H1, H2.  Like hidden_pass_wg (maf_wg.h), with the
// tiles of the two triangular layers dealt by the host-built ownership table: tile cost ranks per wave, balanced over the
// four SIMDs the waves share (the f32 matrix pipe of a SIMD is what bounds a layer).
template <int NW, int PF, bool PROF>
__device__ __forceinline__ void hidden_pass_train(const pmc_maf_t& m, const MafView& w, const float* X, float* H0,
                                                  float* H1, float* H2, int wv, int lane, const int* own,
                                                  long long* pacc, long long& tk) {
    const int q = lane >> 4, p = lane & 15;
    const int nT = m.nT, nXT = m.nXT;
    for (int T = wv; T < nT; T += NW) {
        f32x4 a = bias4(w.b0, 16 * T + 4 * q);
        a = mac_range<PF>(a, w.f0 + (size_t)T * nXT * 64, X, 0, nXT, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = fmaxf(a[r], 0.0f);
        store_rows(H0, T, q, p, a);
    }
    LAPT(15)
    lds_barrier();
    LAPT(14)
    for (int layer = 1; layer <= 2; ++layer) {
        const float* Hin = layer == 1 ? H0 : H1;
        float* Hout = layer == 1 ? H1 : H2;
        const float4* f = layer == 1 ? w.f1 : w.f2;
        const float* b = layer == 1 ? w.b1 : w.b2;
        for (int it = 0; it < TRAIN_OWN_MAX; ++it) {
            const int r = own[it];
            if (r < 0) break;
            const int T = nT - 1 - r;
            f32x4 a = bias4(b, 16 * T + 4 * q);
            a = mac_range<PF>(a, f + (size_t)T * nT * 64, Hin, 0, m.tri_ok ? T + 1 : nT, lane);
            const f32x4 h = rows_of(Hin, T, q, p);
#pragma unroll
            for (int r2 = 0; r2 < 4; ++r2) a[r2] = fmaxf(a[r2] + h[r2], 0.0f);
            store_rows(Hout, T, q, p, a);
        }
        LAPT(13)
        lds_barrier();
        LAPT(14)
    }
}

// out-layer panel of ranks [16c, 16c+16) of a spline flow: output tiles NOUT c .. NOUT c + NOUT - 1 -> P (local tile index).
// keep != NULL: the panel (NOUT tiles of 256 floats, one float4 per lane and tile) is also written there for the
// backward sweep; from != NULL: it is read back from there instead of being multiplied out.
template <int NW, int PF, int NOUT>
__device__ __forceinline__ void rqs_panel_train(const pmc_maf_t& m, const MafView& w, const float* H2, float* P, int c,
                                                int wv, int lane, float* keep = nullptr, const float* from = nullptr) {
    const int q = lane >> 4, p = lane & 15;
    for (int i = wv; i < NOUT; i += NW) {
        const int O = NOUT * c + i;
        if (16 * O >= NOUT * m.D) continue;              // padding rows: never read
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

// UNI 0: affine univariate (MAF), 2 outputs per feature.  UNI = K > 0: K-bin spline (NSF; K = 4, 8, 16), 3 K - 1 outputs.
// One workgroup per row set of 16 (rows 16 (set0 + blockIdx.x) ..., scratch block blockIdx.x).
template <int NW, int PF, bool PROF, int UNI>
__global__ __launch_bounds__(64 * NW) void maf_chain_kernel(pmc_maf_t m, pmc_maf_train_t tr,
                                                                  const float* __restrict__ x,
                                                                  const float* __restrict__ w,
                                                                  const int64_t* __restrict__ idx, float wmul,
                                                                  int64_t n, int64_t set0, int ksplit,
                                                                  long long* __restrict__ prof) {
    constexpr int NOUT = UNI ? RQS_NOUT_OF(UNI) : 2;
    constexpr int KB = UNI ? UNI : 8;                           // bins of the spline instances
    long long pacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tk = TICKT();
    const long long t_begin = tk;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: tile ownership branches stay scalar
    const int q = lane >> 4, p = lane & 15;
    const int D = m.D, Dp = m.Dp, Hp = m.Hp, T = m.T, nT = m.nT, nXT = m.nXT, nOT = m.nOT;
    const int Psz = UNI ? NOUT * 256 : 2 * Dp * 16;       // spline flows: one 16-rank panel at a time
    const int nOeff = UNI ? (NOUT * D + 15) / 16 : min(nOT, (D + 7) / 8);   // output tiles with real rows
    float* A = smem;
    float* B = A + Hp * 16;
    float* Cb = B + Hp * 16;
    float* E = Cb + Hp * 16;
    float* P = E + Hp * 16;                   // [Psz]
    float* Gb = P + Psz;                      // [Dp*16]
    float* CC = Gb + Dp * 16;                 // [16] per-row loss coefficient
    float* RED = CC + 16;                     // [16 * NW]
    float* XB = RED + 16 * NW;       // UNI 1: [Dp*16] the transform's input during its backward sweep
    float* PART = XB + (UNI ? Dp * 16 : 0);   // ksplit: [NW][256] partial tiles of the phases with fewer tiles than waves
    const int* feat_of_rank = m.meta + 8;
    // host-built tables (MAFSpec.train_tables): which hidden tiles a wave owns (by cost rank, balanced over the SIMDs),
    // and where a rank of transform t sits in transform t + 1 / t - 1
    const int* own = tr.tables + wv * TRAIN_OWN_MAX;
    const int* nxt_rank = tr.tables + TRAIN_WAVES * TRAIN_OWN_MAX;
    const int* prv_rank = nxt_rank + T * D;
    const int64_t set = blockIdx.x;
    // what the forward sweep keeps for the backward sweep and both keep for the weight-gradient kernel
    float* xt = tr.xt_scratch + (size_t)set * (T + 1) * Dp * 16;           // input of every transform (+ z)
    float* act = tr.act_scratch + (size_t)set * T * 3 * Hp * 16;            // h0, h1, h2 of every transform
    float* dlt = tr.delta_scratch + (size_t)set * T * 3 * Hp * 16;          // da0, da1, da2 of every transform
    float* par = tr.par_scratch + (size_t)set * T * tr.par_per_transform;   // outputs, then their gradients

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

    float loss_acc = 0.0f;
    {
        const int64_t row0 = (set0 + set) * 16;
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
            float* part = par + (size_t)t * tr.par_per_transform;
            hidden_pass_train<NW, PF, PROF>(m, wvw, Xc, A, B, Cb, wv, lane, own, pacc, tk);
            {                                                // A, B, Cb are contiguous in LDS: one copy
                float4* dst = reinterpret_cast<float4*>(act + (size_t)t * 3 * Hp * 16);
                for (int e = tid; e < 3 * Hp * 4; e += (64 * NW)) dst[e] = reinterpret_cast<const float4*>(A)[e];
            }
            LAPT(2)
            if (UNI == 0) {
                // fewer output tiles than half the waves: a tile's contraction is split over NW / tiles waves (every
                // fragment of the layer in flight at once, the SIMDs evenly loaded), partial tiles meet in LDS
                const int nch = ksplit ? NW / nOeff : 1;
                if (nch > 1) {
                    const int O = wv % nOeff, ch = wv / nOeff;
                    if (ch > 0 && ch < nch) {
                        f32x4 o = {0.f, 0.f, 0.f, 0.f};
                        o = mac_range<PF>(o, wvw.f3 + (size_t)O * nT * 64, Cb, ch * nT / nch, (ch + 1) * nT / nch, lane);
                        reinterpret_cast<float4*>(PART)[(wv - nOeff) * 64 + lane] = make_float4(o[0], o[1], o[2], o[3]);
                    }
                }
                for (int O = wv; O < nOeff; O += NW) {
                    // where this lane's two ranks go in the next transform (requested ahead of the products)
                    int r2[2];
                    for (int s = 0; s < 2; ++s) {
                        const int rank = 8 * O + 2 * q + s;
                        r2[s] = (rank < D && t + 1 < T) ? nxt_rank[t * D + rank] : rank;
                    }
                    f32x4 o = bias4(wvw.b3, 16 * O + 4 * q);
                    o = mac_range<PF>(o, wvw.f3 + (size_t)O * nT * 64, Cb, 0, nch > 1 ? nT / nch : nT, lane);
                    if (nch > 1) {
                        lds_barrier();
                        for (int ch = 1; ch < nch; ++ch) {          // chunk order: a fixed order of additions
                            const float4 v = reinterpret_cast<const float4*>(PART)[(ch * nOeff + O - nOeff) * 64 + lane];
                            o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w;
                        }
                    }
                    store_rows(part, O, q, p, o);            // (shift, raw): the backward sweep reads them back
                    for (int s = 0; s < 2; ++s) {
                        const int rank = 8 * O + 2 * q + s;
                        if (rank < D) {
                            const float shift = s ? o[2] : o[0];
                            const float ls = soft_ls(s ? o[3] : o[1]);
                            const float y = Xc[lidx(rank, p)] * expf(ls) + shift;
                            Xn[lidx(r2[s], p)] = y;          // the next transform reads its input by its own rank order
                            xtn[lidx(r2[s], p)] = y;
                            ladj += ls;
                        }
                    }
                }
                if (nch > 1 && wv >= nOeff) lds_barrier();       // (the waves without a tile meet the owners' barrier)
            } else {
                for (int c = 0; c < nXT; ++c) {
                    rqs_panel_train<NW, PF, NOUT>(m, wvw, Cb, P, c, wv, lane, part + (size_t)c * NOUT * 256);
                    lds_barrier();
                    LAPT(11)
                    for (int e = tid; e < 256; e += (64 * NW)) {
                        const int rr = e >> 4, pp = e & 15, rank = 16 * c + rr;
                        if (rank < D) {
                            float phi[NOUT];
#pragma unroll
                            for (int j = 0; j < NOUT; ++j) phi[j] = P[lidx(NOUT * rr + j, pp)];
                            float y, l;
                            rqs_forward_t<KB>(phi, Xc[lidx(rank, pp)], y, l);
                            const int r2 = (t + 1 < T) ? nxt_rank[t * D + rank] : rank;
                            Xn[lidx(r2, pp)] = y;
                            xtn[lidx(r2, pp)] = y;
                            ladj += l;
                        }
                    }
                    LAPT(12)
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
            __syncthreads();                             // full barrier: the scratch written above is read below
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
            const TrainView tv = train_view(m, tr, t);
            const float4* xsrc = reinterpret_cast<const float4*>(xt + (size_t)t * Dp * 16);
            float* part = par + (size_t)t * tr.par_per_transform;
            float* dl = dlt + (size_t)t * 3 * Hp * 16;
            const float4* asrc = reinterpret_cast<const float4*>(act + (size_t)t * 3 * Hp * 16);
            if (UNI == 0) {
                // x_t -> E, (h0, h1, h2) -> A, B, C, (shift, raw) -> P: everything the forward sweep kept
                for (int e = tid; e < Dp * 4; e += (64 * NW)) reinterpret_cast<float4*>(E)[e] = xsrc[e];
                for (int e = tid; e < nOeff * 64; e += (64 * NW))
                    reinterpret_cast<float4*>(P)[e] = reinterpret_cast<const float4*>(part)[e];
                if (t < T - 1)                               // (the last transform's activations never left LDS)
                    for (int e = tid; e < 3 * Hp * 4; e += (64 * NW)) reinterpret_cast<float4*>(A)[e] = asrc[e];
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
                // ---- layer 3: da2 = relu'(h2) . W3^T dP -> E ; dP -> scratch (dW3, db3)
                for (int e = tid; e < nOeff * 64; e += (64 * NW))
                    reinterpret_cast<float4*>(part)[e] = reinterpret_cast<const float4*>(P)[e];
                for (int K = wv; K < nT; K += NW) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f};
                    a = mac_range<PF>(a, tv.f3T + (size_t)K * nOT * 64, P, 0, nOeff, lane);
                    a = relu_gate(a, Cb, K, q, p);
                    store_rows(E, K, q, p, a);
                    store_rows(dl + 2 * Hp * 16, K, q, p, a);
                }
                PHASE_END(6)
            } else {
                const MafView wvw = maf_view(m, t);
                // x_t -> XB (kept for the whole spline sweep), E <- 0 (accumulates W3^T dP over the panels)
                for (int e = tid; e < Dp * 4; e += (64 * NW)) reinterpret_cast<float4*>(XB)[e] = xsrc[e];
                for (int e = tid; e < Hp * 4; e += (64 * NW))
                    reinterpret_cast<float4*>(E)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t < T - 1)
                    for (int e = tid; e < 3 * Hp * 4; e += (64 * NW)) reinterpret_cast<float4*>(A)[e] = asrc[e];
                lds_barrier();
                LAPT(4)
                for (int c = 0; c < nXT; ++c) {
                    const int O0 = NOUT * c;                         // first output tile of the panel
                    const int nO = min(NOUT, nOeff - O0);            // its tiles with real rows
                    float* pc = part + (size_t)c * NOUT * 256;
                    rqs_panel_train<NW, PF, NOUT>(m, wvw, Cb, P, c, wv, lane, nullptr, pc);
                    PHASE_END(4)
                    // spline backward in place: P -> dP, G -> direct dL/dx term
                    for (int e = tid; e < 256; e += (64 * NW)) {
                        const int rr = e >> 4, pp = e & 15, rank = 16 * c + rr;
                        if (rank < D) {
                            float phi[NOUT], dphi[NOUT];
#pragma unroll
                            for (int j = 0; j < NOUT; ++j) phi[j] = P[lidx(NOUT * rr + j, pp)];
                            float gx;
                            rqs_backward_t<KB>(phi, XB[lidx(rank, pp)], Gb[lidx(rank, pp)], -CC[pp], dphi, gx);
#pragma unroll
                            for (int j = 0; j < NOUT; ++j) P[lidx(NOUT * rr + j, pp)] = dphi[j];
                            Gb[lidx(rank, pp)] = gx;
                        }
                    }
                    PHASE_END(5)
                    // W3^T dP of this panel into E (every wave owns whole tiles of E); dP -> scratch (dW3, db3)
                    for (int e = tid; e < nO * 64; e += (64 * NW))
                        reinterpret_cast<float4*>(pc)[e] = reinterpret_cast<const float4*>(P)[e];
                    for (int K = wv; K < nT; K += NW) {
                        f32x4 a = rows_of(E, K, q, p);
                        a = mac_range<PF>(a, tv.f3T + ((size_t)K * nOT + O0) * 64, P, 0, nO, lane);
                        store_rows(E, K, q, p, a);
                    }
                    PHASE_END(6)
                }
                // da2 = relu'(h2) . (W3^T dP)
                for (int K = wv; K < nT; K += NW) {
                    f32x4 a = relu_gate(rows_of(E, K, q, p), Cb, K, q, p);
                    store_rows(E, K, q, p, a);
                    store_rows(dl + 2 * Hp * 16, K, q, p, a);
                }
                PHASE_END(6)
            }
            // ---- layer 2: da1 = relu'(h1) . (da2 + W2^T da2) -> C
            for (int it = 0; it < TRAIN_OWN_MAX; ++it) {           // the tiles this wave owns, most expensive first
                const int Ti = own[it];
                if (Ti < 0) break;
                f32x4 a = rows_of(E, Ti, q, p);
                a = mac_range<PF>(a, tv.f2T + (size_t)Ti * nT * 64, E, (m.tri_ok ? Ti : 0), nT, lane);
                a = relu_gate(a, B, Ti, q, p);
                store_rows(Cb, Ti, q, p, a);
                store_rows(dl + Hp * 16, Ti, q, p, a);
            }
            PHASE_END(7)
            // ---- layer 1: da0 = relu'(h0) . (da1 + W1^T da1) -> E
            for (int it = 0; it < TRAIN_OWN_MAX; ++it) {
                const int Ti = own[it];
                if (Ti < 0) break;
                f32x4 a = rows_of(Cb, Ti, q, p);
                a = mac_range<PF>(a, tv.f1T + (size_t)Ti * nT * 64, Cb, (m.tri_ok ? Ti : 0), nT, lane);
                a = relu_gate(a, A, Ti, q, p);
                store_rows(E, Ti, q, p, a);
                store_rows(dl, Ti, q, p, a);
            }
            PHASE_END(8)
            // ---- layer 0: dx = direct + W0^T da0 -> P
            if (t > 0) {
                const int nch = (ksplit && 2 * nXT <= NW) ? min(nT, NW / nXT) : 1;
                if (nch > 1) {
                    const int Xi = wv % nXT, ch = wv / nXT;
                    if (ch > 0 && ch < nch) {
                        f32x4 a = {0.f, 0.f, 0.f, 0.f};
                        a = mac_range<PF>(a, tv.f0T + (size_t)Xi * nT * 64, E, ch * nT / nch, (ch + 1) * nT / nch, lane);
                        reinterpret_cast<float4*>(PART)[(wv - nXT) * 64 + lane] = make_float4(a[0], a[1], a[2], a[3]);
                    }
                }
                for (int Xi = wv; Xi < nXT; Xi += NW) {
                    f32x4 a = rows_of(Gb, Xi, q, p);
                    a = mac_range<PF>(a, tv.f0T + (size_t)Xi * nT * 64, E, 0, nch > 1 ? nT / nch : nT, lane);
                    if (nch > 1) {
                        lds_barrier();
                        for (int ch = 1; ch < nch; ++ch) {
                            const float4 v = reinterpret_cast<const float4*>(PART)[(ch * nXT + Xi - nXT) * 64 + lane];
                            a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
                        }
                    }
                    store_rows(P, Xi, q, p, a);
                }
                if (nch > 1 && wv >= nXT) lds_barrier();
                PHASE_END(9)
                // re-rank for transform t-1 (its output order)
                for (int e = tid; e < Dp * 16; e += (64 * NW)) {
                    const int r = e >> 4, pp = e & 15;
                    if (r < D) Gb[lidx(prv_rank[t * D + r], pp)] = P[lidx(r, pp)];
                }
                PHASE_END(9)
            }
        }
    }
    if (tid == 0) tr.loss_partial[set] = loss_acc;
    if (PROF && lane == 0) {
        pacc[0] = TICKT() - t_begin;
        long long* o = prof + ((size_t)blockIdx.x * NW + wv) * 16;
        for (int i = 0; i < 16; ++i) o[i] = pacc[i];
    }
}

// Weight- and bias-gradient tiles over the whole batch (header comment, part 2).  One workgroup per job
// (MAFSpec.train_jobs), int32 [8]: {kind_a, off_a, kind_b, off_b, gmap offset of the tile | -1, gmap offset of the 16
// bias entries | -1, 0, 0} (kind_a = -1: no-op); kind 0: xt_scratch, 1: act_scratch, 2: delta_scratch, 3: par_scratch, offsets in floats
// inside a row set's block.  A = delta tile (out), B = activation tile (in):
//     dW[out 16 To + i][in 16 Ti + j] = sum_sets sum_p A[i][p] B[j][p],    db[16 To + i] = sum_sets sum_p A[i][p].
// accumulate: the batch comes in chunks of at most max_sets row sets, later chunks add to the gradient in place.
#define DW_WAVES 4
#define DW_AHEAD 8
__global__ __launch_bounds__(64 * DW_WAVES) void maf_dw_kernel(pmc_maf_t m, pmc_maf_train_t tr, int nsets, int accumulate,
                                                               float* __restrict__ grad, float* __restrict__ loss) {
    __shared__ float red[DW_WAVES - 1][64 * 4];
    __shared__ float redb[DW_WAVES - 1][16];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int* job = tr.jobs + (size_t)blockIdx.x * 8;
    const int ka = job[0], oa = job[1], kb = job[2], ob = job[3], gw = job[4], gbias = job[5];
    if (ka < 0) {                                               // padding of the XCD placement (MAFSpec.train_jobs)
        if (threadIdx.x == 0) tr.sq_partial[blockIdx.x] = 0.0f;
        return;
    }
    const int64_t s_xt = (int64_t)(m.T + 1) * m.Dp * 16, s_act = (int64_t)m.T * 3 * m.Hp * 16,
                  s_par = (int64_t)m.T * tr.par_per_transform;
    const float* base_a = ka == 0 ? tr.xt_scratch : ka == 1 ? tr.act_scratch : ka == 2 ? tr.delta_scratch : tr.par_scratch;
    const float* base_b = kb == 0 ? tr.xt_scratch : kb == 1 ? tr.act_scratch : kb == 2 ? tr.delta_scratch : tr.par_scratch;
    const int64_t sa = ka == 0 ? s_xt : ka == 3 ? s_par : s_act, sb = kb == 0 ? s_xt : kb == 3 ? s_par : s_act;
    // operand element [i = lane & 15][p = 4 c + (lane >> 4)] of a tile stored by store_rows / lidx
    const int i = lane & 15, kq = lane >> 4;
    const int lo = ((i & 3) << 6) + (i >> 2) + (kq << 2);
    const float* pa = base_a + oa + lo;
    const float* pb = base_b + (gw >= 0 ? ob : 0) + lo;
    const int per = (nsets + DW_WAVES - 1) / DW_WAVES;
    const int s0 = wv * per, s1 = min(nsets, s0 + per);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float bs = 0.0f;
    for (int sg = s0; sg < s1; sg += DW_AHEAD) {
        float a[DW_AHEAD][4], b[DW_AHEAD][4];
#pragma unroll
        for (int u = 0; u < DW_AHEAD; ++u) {
            const int s = min(sg + u, s1 - 1);                // (clamped: the loads stay branch free)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                a[u][c] = pa[(int64_t)s * sa + 16 * c];
                b[u][c] = gw >= 0 ? pb[(int64_t)s * sb + 16 * c] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < DW_AHEAD; ++u) {
            if (sg + u < s1) {
                if (u & 1) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc1 = MFMA(a[u][c], b[u][c], acc1);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc0 = MFMA(a[u][c], b[u][c], acc0);
                }
                bs += (a[u][0] + a[u][1]) + (a[u][2] + a[u][3]);
            }
        }
    }
    f32x4 acc = acc0 + acc1;
    bs += __shfl_xor(bs, 16);
    bs += __shfl_xor(bs, 32);                                   // lanes 0..15: this wave's share of db[16 To + lane]
    if (wv > 0) {
        reinterpret_cast<float4*>(red[wv - 1])[lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        if (lane < 16) redb[wv - 1][lane] = bs;
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int k = 0; k < DW_WAVES - 1; ++k) {                // wave order: a fixed order of additions
            const float4 v = reinterpret_cast<const float4*>(red[k])[lane];
            acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
            if (lane < 16) bs += redb[k][lane];
        }
        float sq = 0.0f;
        if (gw >= 0) {
            const int4 g = reinterpret_cast<const int4*>(tr.gmap + gw)[lane];
            const int gi[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (gi[r] >= 0) {
                    const float v = accumulate ? grad[gi[r]] + acc[r] : acc[r];
                    grad[gi[r]] = v;
                    sq += v * v;
                }
            }
        }
        if (gbias >= 0 && lane < 16) {
            const int g = tr.gmap[gbias + lane];
            if (g >= 0) {
                const float v = accumulate ? grad[g] + bs : bs;
                grad[g] = v;
                sq += v * v;
            }
        }
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        if (lane == 0) {
            tr.sq_partial[blockIdx.x] = sq;
            if (blockIdx.x == 0 && loss) {
                float s = 0.0f;
                for (int b = 0; b < nsets; ++b) s += tr.loss_partial[b];
                *loss += s;
            }
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
                                                    float* __restrict__ img_a, int n_a, float* __restrict__ img_b,
                                                    float* __restrict__ snap) {
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
        if (snap) snap[e] = pv;                           // (pmc_adamw_t.snapshot: the parameters behind an epoch's last step)
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
static size_t train_lds_bytes(const pmc_maf_t& m, bool ksplit) {
    const size_t part = ksplit ? (size_t)TRAIN_WAVES * 256 : 0;
    if (m.n_out != 2)
        return (size_t)(4 * m.Hp * 16 + m.n_out * 256 + 2 * m.Dp * 16 + 16 + 16 * TRAIN_WAVES + part) * sizeof(float);
    return (size_t)(4 * m.Hp * 16 + 2 * m.Dp * 16 + m.Dp * 16 + 16 + 16 * TRAIN_WAVES + part) * sizeof(float);   // (sized for 16 waves)
}

static int train_check(const pmc_maf_t* m, const pmc_maf_train_t* tr, const char* who) {
    if (!m || !tr || !tr->packedT || !tr->gmap || !tr->jobs || tr->n_jobs < 1 || tr->max_sets < 1 || !tr->xt_scratch ||
        !tr->act_scratch || !tr->delta_scratch || !tr->par_scratch || !tr->loss_partial || !tr->sq_partial ||
        tr->n_sq_partial < tr->n_jobs || !tr->tables || tr->table_waves != train_waves_of(*m) ||
        tr->par_per_transform < (int64_t)(m->n_out != 2 ? m->nXT * m->n_out : m->nOT) * 256)
        return pmc_fail((std::string(who) + ": incomplete training image").c_str());
    return 0;
}

static int launch_lossgrad(const pmc_maf_t* m, const pmc_maf_train_t* tr, const float* x, const float* w,
                           const int64_t* idx, float wmul, float* grad, float* loss, int64_t n, hipStream_t st,
                           long long* prof = nullptr) {
    // partial-tile staging for the phases with fewer tiles than waves, where the workgroup's LDS has room for it
    const bool ksplit = train_lds_bytes(*m, true) <= 64 * 1024;
    const size_t lds = train_lds_bytes(*m, ksplit);
    if (lds > 160 * 1024) return pmc_fail("pmc_maf_loss_grad: flow too wide for the 160 KB LDS of one workgroup");
    const bool narrow = train_waves_of(*m) == TRAIN_WAVES_NARROW;
    const int64_t nsets = (n + 15) / 16;
    // a batch of more row sets than the scratch arrays hold comes in chunks: chain + weight gradients per chunk, the
    // later chunks adding to the gradient in place (the same order of additions whatever the rows are)
    for (int64_t set0 = 0; set0 < nsets; set0 += tr->max_sets) {
        const int n_wg = (int)(nsets - set0 < tr->max_sets ? nsets - set0 : tr->max_sets);
#define LG(NWV, PFV, PR, UN)                                                                                       \
    {                                                                                                              \
        static size_t lds_set = 0;                                                                                 \
        if (lds > 48 * 1024 && lds > lds_set) {                                                                    \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(maf_chain_kernel<NWV, PFV, PR, UN>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
            if (e != hipSuccess) return pmc_fail_hip(e, "hipFuncSetAttribute(maf_chain_kernel)");                  \
            lds_set = lds;                                                                                         \
        }                                                                                                          \
        hipLaunchKernelGGL((maf_chain_kernel<NWV, PFV, PR, UN>), dim3((unsigned)n_wg), dim3(64 * NWV), lds, st, *m, *tr, \
                           x, w, idx, wmul, n, set0, ksplit ? 1 : 0, prof);                                        \
    }
#define LGK(PR, K)                                                                                                 \
    if (m->n_out == RQS_NOUT_OF(K)) {                                                                              \
        if (narrow) LG(TRAIN_WAVES_NARROW, TRAIN_PF_NARROW, PR, K) else LG(TRAIN_WAVES, TRAIN_PF, PR, K)           \
    } else
#ifdef PMC_DEBUG_HOOKS
        if (prof) {
            LGK(true, 8) LGK(true, 4) LGK(true, 16) LG(TRAIN_WAVES, TRAIN_PF, true, 0)
        } else
#endif
        { LGK(false, 8) LGK(false, 4) LGK(false, 16) LG(TRAIN_WAVES, TRAIN_PF, false, 0) }
#undef LGK
#undef LG
        hipLaunchKernelGGL(maf_dw_kernel, dim3((unsigned)tr->n_jobs), dim3(64 * DW_WAVES), 0, st, *m, *tr, n_wg,
                           set0 > 0 ? 1 : 0, grad, loss);
    }
    return pmc_check_launch("maf_chain_kernel");
}

extern "C" int pmc_maf_loss_grad(const pmc_maf_t* m, const pmc_maf_train_t* tr, const float* x, const float* w,
                                 const int64_t* idx, float wmul, float* grad, float* loss, int64_t n, void* stream) {
    if (train_check(m, tr, "pmc_maf_loss_grad")) return 1;
    if (!x || !grad || !loss || n < 0) return pmc_fail("pmc_maf_loss_grad: bad argument");
    if (n == 0) return 0;
    return launch_lossgrad(m, tr, x, w, idx, wmul, grad, loss, n, (hipStream_t)stream);
}

#ifdef PMC_DEBUG_HOOKS
// in-kernel cycle profile of the chain kernel (scripts/profile_train.py; DEBUG_HOOKS builds only): prof i64 [sets][waves][16]
extern "C" int pmc_debug_lossgrad_profile(const pmc_maf_t* m, const pmc_maf_train_t* tr, const float* x, float* grad,
                                          float* loss, int64_t n, long long* prof, void* stream) {
    if (train_check(m, tr, "pmc_debug_lossgrad_profile")) return 1;
    return launch_lossgrad(m, tr, x, nullptr, nullptr, 1000.0f, grad, loss, n, (hipStream_t)stream, prof);
}
#endif
// wavefronts per training workgroup for this flow: what MAFSpec.train_tables builds the ownership table for
extern "C" int pmc_maf_train_waves(const pmc_maf_t* m) { return m ? train_waves_of(*m) : TRAIN_WAVES; }

// validation loss of a whole pass (flow.py:327-348) from the rows' log-densities: per batch b of `bs` rows
// loss += sum_i -(logp_i * c_i), c_i = 1 or w[idx_i] * wmul / sum_j w[idx_j] over the batch.  ONE block walks the batches
// in order: a fixed order of additions.
__global__ __launch_bounds__(256) void epoch_nll_kernel(const float* __restrict__ logp, const float* __restrict__ w,
                                                        const int64_t* __restrict__ idx, float wmul,
                                                        float* __restrict__ loss, int64_t n, int64_t bs) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    float total = *loss;
    for (int64_t b0 = 0; b0 < n; b0 += bs) {
        const int64_t nb = n - b0 < bs ? n - b0 : bs;
        float scale = 1.0f;
        if (w) {
            float s = 0.0f;
            for (int64_t i = tid; i < nb; i += 256) s += w[idx ? idx[b0 + i] : b0 + i];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            __syncthreads();
            if ((tid & 63) == 0) red[tid >> 6] = s;
            __syncthreads();
            scale = wmul / ((red[0] + red[1]) + (red[2] + red[3]));
        }
        float s = 0.0f;
        for (int64_t i = tid; i < nb; i += 256) s += -(logp[b0 + i] * (w ? w[idx ? idx[b0 + i] : b0 + i] * scale : 1.0f));
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        total += (red[0] + red[1]) + (red[2] + red[3]);
    }
    if (tid == 0) *loss = total;
}

// One pass over a validation set in batches (flow.py:327-348), everything enqueued by one call: ONE forward launch over all
// n rows (a row's log-density does not depend on its batch), one launch for the batches' weighted sums.
extern "C" int pmc_maf_valid_epoch(const pmc_maf_t* m, const float* x, const float* w, const int64_t* perm, int64_t n,
                                   int64_t batch_size, float* logp_scratch, float* loss, void* stream) {
    if (!m || !x || !logp_scratch || !loss || n < 0 || batch_size < 1) return pmc_fail("pmc_maf_valid_epoch: bad argument");
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (int rc = pmc_launch_forward_wg(m, x, nullptr, nullptr, logp_scratch, n, st, perm)) return rc;
    hipLaunchKernelGGL(epoch_nll_kernel, dim3(1), dim3(256), 0, st, (const float*)logp_scratch, w, perm, 1000.0f, loss, n,
                       batch_size);
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
                         float* img_a = nullptr, int n_a = 0, float* img_b = nullptr, float* snap = nullptr) {
    int64_t grid = (n + 255) / 256; if (grid > 512) grid = 512;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)grid), dim3(256), 0, st, params, grad, m1, m2, n, (float)lr,
                       (float)beta1, (float)beta2, (float)eps, (float)wd, (float)max_norm, (float)bc1, (float)bc2,
                       sq_part, n_part, sc_ptr, sc_dst, img_a, n_a, img_b, snap);
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

extern "C" int pmc_maf_train_epoch_gated(const pmc_maf_t* m, const pmc_maf_train_t* tr, pmc_adamw_t* opt, const float* x,
                                         const float* w, const int64_t* perm, int64_t n, int64_t batch_size, float* loss,
                                         void* gate_event, void* stream);

extern "C" int pmc_maf_train_epoch(const pmc_maf_t* m, const pmc_maf_train_t* tr, pmc_adamw_t* opt, const float* x,
                                   const float* w, const int64_t* perm, int64_t n, int64_t batch_size, float* loss,
                                   void* stream) {
    return pmc_maf_train_epoch_gated(m, tr, opt, x, w, perm, n, batch_size, loss, nullptr, stream);
}

// gate_event: the first OPTIMIZER STEP of the epoch waits for it (the loss / gradient launches in front of it only read
// the parameters): the validation pass of the previous epoch, running on another stream on the compute units the chain
// kernel leaves idle, reads the kernel images this step rewrites.
extern "C" int pmc_maf_train_epoch_gated(const pmc_maf_t* m, const pmc_maf_train_t* tr, pmc_adamw_t* opt, const float* x,
                                         const float* w, const int64_t* perm, int64_t n, int64_t batch_size, float* loss,
                                         void* gate_event, void* stream) {
    if (train_check(m, tr, "pmc_maf_train_epoch")) return 1;
    if (!opt || !opt->params || !opt->grad || !opt->exp_avg || !opt->exp_avg_sq || !opt->pack_idx || !opt->packed ||
        !opt->packT_idx || !opt->packedT || opt->n_params <= 0 || !x || !loss || n < 0 || batch_size < 1)
        return pmc_fail("pmc_maf_train_epoch: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int n_part = tr->n_jobs;                      // maf_dw_kernel's block sums of squares
    for (int64_t b0 = 0; b0 < n; b0 += batch_size) {
        const int64_t nb = (n - b0 < batch_size) ? n - b0 : batch_size;
        // a batch is rows perm[b0 .. b0+nb) of x, or rows b0 .. b0+nb when perm == NULL
        const float* xb = perm ? x : x + b0 * m->D;
        const float* wb = (w && !perm) ? w + b0 : w;
        if (launch_lossgrad(m, tr, xb, wb, perm ? perm + b0 : nullptr, 1000.0f, opt->grad, loss, nb, st)) return 1;
        if (gate_event && b0 == 0) {
            const hipError_t e = hipStreamWaitEvent(st, (hipEvent_t)gate_event, 0);
            if (e != hipSuccess) return pmc_fail_hip(e, "hipStreamWaitEvent(pmc_maf_train_epoch_gated)");
        }
        opt->step += 1;
        const bool scatter = opt->scatter_ptr && opt->scatter_dst && opt->n_packed < 0x7fffffffLL;
        launch_adamw(opt->params, opt->grad, opt->exp_avg, opt->exp_avg_sq, opt->n_params, opt->lr, opt->beta1,
                     opt->beta2, opt->eps, opt->weight_decay, opt->max_norm, opt->step, tr->sq_partial, n_part, st,
                     scatter ? opt->scatter_ptr : nullptr, opt->scatter_dst, opt->packed, (int)opt->n_packed,
                     opt->packedT, (b0 + batch_size >= n) ? opt->snapshot : nullptr);
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

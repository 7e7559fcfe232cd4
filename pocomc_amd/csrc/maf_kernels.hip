// MAF (masked autoregressive flow) kernels for gfx950 (MI355X, CDNA4).
//
// Replaces, on the device, what pocomc/flow.py:99-163 gets from zuko:
//   forward  (data -> latent, ladj)          flow.py:99-114
//   inverse  (latent -> data, ladj)          flow.py:116-132   <- every MCMC step (mcmc.py:88)
//   log_prob                                 flow.py:134-147
//
// Execution model: one wavefront (64 lanes) owns 16 particles.  All matrix work
// is exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), computed transposed:
//     D[out row i][particle j] += sum_k  W[out i][in k] * act[in k][particle j]
// so the A operand is a weight fragment (pre-packed on the host side in lane
// order, one coalesced 1 KiB float4 load feeds 4 MFMAs) and the B operand is a
// slice of the activations, which live in wave-private LDS in a layout where
// the B operands of 4 consecutive K-chunks are one ds_read_b128:
//     idx(row r, particle p) = (r>>4)*256 + (r&3)*64 + p*4 + ((r>>2)&3)
// Lane l = (q = l>>4, p = l&15) of an accumulator holds rows 4q..4q+3 of the
// 16-row tile for particle p.
//
// The inverse is NOT the reference's D fixed-point passes: hidden units are
// sorted by autoregressive degree on the host (maf_spec.py), so the masked
// weights are block lower-triangular and the inverse is a single sweep over the
// degree groups (a blocked triangular solve): per hidden tile one left-looking
// "burst" against everything already final, then per degree group a short
// dependent chain  h0 -> h1 -> h2 -> (shift, raw) -> x_rank -> rank-1 update.
// Work = one masked forward pass instead of D+1 dense ones.  The D-pass
// algorithm is kept (mode NAIVE) as the on-device cross-check and for layouts
// whose degree groups exceed one tile.

#include "maf_common.h"
#include "propose_body.h"

// ---------------------------------------------------------------------------
// Dense pass of the hyper-network of one transform for 16 particles:
// in: xin (LDS, by rank).  out: natural (shift, raw) accumulators are consumed
// by the callback-free epilogue below.  H0/H1/H2 are scratch.
// mode 0: forward  y = x*exp(ls)+shift      (xin = x, writes y to xout)
// mode 1: inverse pass x' = (y-shift)/exp(ls) (xin = current x, yin = y)
// ---------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ float maf_dense_pass(const pmc_maf_t& m, const MafView& w,
                                                const float* xin, const float* yin, float* xout,
                                                float* H0, float* H1, float* H2,
                                                int lane, bool want_ladj) {
    const int q = lane >> 4, p = lane & 15;
    const int nT = m.nT, nOT = m.nOT, D = m.D;
    maf_hidden_pass(m, w, xin, H0, H1, H2, lane);
    float ladj = 0.0f;
    for (int O = 0; O < nOT; ++O) {
        if (8 * O >= D) break;
        f32x4 o = bias4(w.b3, 16 * O + 4 * q);
        o = mac_range(o, w.f3 + (size_t)O * nT * 64, H2, 0, nT, lane);
        for (int s = 0; s < 2; ++s) {
            const int rank = 8 * O + 2 * q + s;
            if (rank < D) {
                const float shift = s ? o[2] : o[0];
                const float ls = soft_ls(s ? o[3] : o[1]);
                if (MODE == 0) {
                    const float x = xin[lidx(rank, p)];
                    xout[lidx(rank, p)] = x * expf(ls) + shift;
                } else {
                    const float y = yin[lidx(rank, p)];
                    xout[lidx(rank, p)] = (y - shift) / expf(ls);
                }
                if (want_ladj) ladj += ls;
            }
        }
    }
    __syncthreads();
    return ladj;
}

// ---------------------------------------------------------------------------
// forward (MODE 0) and naive D-pass inverse (MODE 1)
// ---------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(64) void maf_dense_kernel(pmc_maf_t m, const float* __restrict__ in,
                                                       float* __restrict__ out, float* __restrict__ ladj_out,
                                                       float* __restrict__ logprob_out, int64_t n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int p = lane & 15;
    const int64_t row0 = (int64_t)blockIdx.x * 16;
    const int D = m.D, Dp = m.Dp, Hp = m.Hp, T = m.T;
    float* A = smem;                 // current values by rank
    float* B = A + Dp * 16;          // next values
    float* C = B + Dp * 16;          // (inverse) iterate
    float* H0 = C + Dp * 16;
    float* H1 = H0 + Hp * 16;
    float* H2 = H1 + Hp * 16;
    const int* feat_of_rank = m.meta + 8;
    const int* rank_of_feat = m.meta + 8 + T * D;

    float ladj = 0.0f;
    if (MODE == 0) {
        load_rows(A, in, row0, n, D, Dp, feat_of_rank, lane);
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            const MafView w = maf_view(m, t);
            ladj += maf_dense_pass<0>(m, w, A, nullptr, B, H0, H1, H2, lane, true);
            const bool last = (t == T - 1);
            rerank_or_store(B, A, out, row0, n, D, Dp, feat_of_rank + t * D,
                            last ? nullptr : rank_of_feat + (t + 1) * D, lane);
            __syncthreads();
            if (last && logprob_out) {
                // base N(0,I) log-density of z (flow.py:147 -> zuko DiagNormal)
                float ss = 0.0f;
                for (int r = (lane >> 4); r < D; r += 4) { const float z = B[lidx(r, p)]; ss += z * z; }
                ss = quad_sum(ss);
                const float l = quad_sum(ladj);
                if (lane < 16 && row0 + p < n)
                    logprob_out[row0 + p] = (-0.5f * ss - 0.9189385332046727f * (float)D) + l;
            }
        }
        ladj = quad_sum(ladj);
        if (ladj_out && lane < 16 && row0 + p < n) ladj_out[row0 + p] = ladj;
    } else {
        load_rows(A, in, row0, n, D, Dp, feat_of_rank + (T - 1) * D, lane);
        __syncthreads();
        for (int t = T - 1; t >= 0; --t) {
            const MafView w = maf_view(m, t);
            for (int e = lane; e < Dp * 16; e += 64) C[e] = 0.0f;
            __syncthreads();
            float* cur = C; float* nxt = B;
            for (int pass = 0; pass < D; ++pass) {      // zuko: passes = features
                maf_dense_pass<1>(m, w, cur, A, nxt, H0, H1, H2, lane, false);
                float* tmp = cur; cur = nxt; nxt = tmp;
                for (int e = lane; e < (Dp - D) * 16; e += 64) cur[lidx(D + (e >> 4), e & 15)] = 0.0f;
                __syncthreads();
            }
            // one more pass for the log-determinant (zuko inv.call_and_ladj)
            ladj -= maf_dense_pass<1>(m, w, cur, A, nxt, H0, H1, H2, lane, true);
            const bool last = (t == 0);
            rerank_or_store(cur, A, out, row0, n, D, Dp, feat_of_rank + t * D,
                            last ? nullptr : rank_of_feat + (t - 1) * D, lane);
            __syncthreads();
        }
        ladj = quad_sum(ladj);
        if (ladj_out && lane < 16 && row0 + p < n) ladj_out[row0 + p] = ladj;
    }
}

// ---------------------------------------------------------------------------
// packing: packed[i] = idx[i] >= 0 ? flat[idx[i]] : 0
// ---------------------------------------------------------------------------
__global__ void maf_pack_kernel(const float* __restrict__ flat, const int32_t* __restrict__ idx,
                                float* __restrict__ packed, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t j = idx[i];
        packed[i] = j >= 0 ? flat[j] : 0.0f;
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static size_t maf_lds_bytes(const pmc_maf_t& m, int n_rank_arrays) {
    return (size_t)(n_rank_arrays * m.Dp * 16 + 3 * m.Hp * 16) * sizeof(float);
}

template <typename K>
static int set_lds(K kernel, size_t bytes) {
    if (bytes > 160 * 1024) return pmc_fail("MAF too wide for one wave's LDS budget (160 KiB)");
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return pmc_fail_hip(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    return 0;
}

extern "C" int pmc_maf_pack(const float* flat, const int32_t* pack_idx, float* packed, int64_t n_packed,
                            void* stream) {
    if (!flat || !pack_idx || !packed || n_packed <= 0) return pmc_fail("pmc_maf_pack: bad argument");
    const int block = 256;
    int64_t grid = (n_packed + block - 1) / block;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(maf_pack_kernel, dim3((unsigned)grid), dim3(block), 0, (hipStream_t)stream,
                       flat, pack_idx, packed, n_packed);
    return pmc_check_launch("maf_pack_kernel");
}

static int check_maf(const pmc_maf_t* m) {
    if (!m || !m->packed || !m->meta) return pmc_fail("pmc_maf: null descriptor field");
    if (m->D < 2 || m->T < 1 || (m->Hp & 15) || (m->Dp & 15) || m->nT * 16 != m->Hp ||
        m->nXT * 16 != m->Dp || (m->n_out != 2 && m->n_out != 11 && m->n_out != 23 && m->n_out != 47) || m->nOT * 16 != m->n_out * m->Dp)
        return pmc_fail("pmc_maf: inconsistent descriptor");
    return 0;
}

extern "C" int pmc_maf_forward(const pmc_maf_t* m, const float* x, float* z, float* ladj, float* log_prob,
                               int64_t n, void* stream) {
    if (int e = check_maf(m)) return e;
    if (n == 0) return 0;
    if (!x || !z || n < 0) return pmc_fail("pmc_maf_forward: bad argument");
    return pmc_launch_forward_wg(m, x, z, ladj, log_prob, n, (hipStream_t)stream);
}

#ifdef PMC_DEBUG_HOOKS
// the lone-wave forward (one wavefront per 16 rows), kept as a cross-check of the workgroup kernel
extern "C" int pmc_debug_forward_lone_wave(const pmc_maf_t* m, const float* x, float* z, float* ladj, float* log_prob,
                                           int64_t n, void* stream) {
    if (int e = check_maf(m)) return e;
    if (n == 0) return 0;
    const size_t lds = maf_lds_bytes(*m, 3);
    if (int e = set_lds(maf_dense_kernel<0>, lds)) return e;
    hipLaunchKernelGGL(maf_dense_kernel<0>, dim3((unsigned)((n + 15) / 16)), dim3(64), lds, (hipStream_t)stream,
                       *m, x, z, ladj, log_prob, n);
    return pmc_check_launch("maf_dense_kernel<forward>");
}
#endif

extern "C" int pmc_maf_inverse(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n,
                               int algo, void* stream) {
    if (int e = check_maf(m)) return e;
    if (n == 0) return 0;
    if (!z || !x || n < 0) return pmc_fail("pmc_maf_inverse: bad argument");
    if (m->n_out != 2) {
        // spline flows: triangular sweep, or the D-pass algorithm of the reference (zuko) as cross-check
        // and for layouts whose degree groups exceed a tile
        // (the sweeps are built for the reference's 8 bins; other bin counts take zuko's own D-pass algorithm)
        if (algo == PMC_INVERSE_AUTO) algo = (m->tri_ok && m->n_out == 23) ? PMC_INVERSE_TRIANGULAR : PMC_INVERSE_NAIVE;
        if (algo == PMC_INVERSE_TRIANGULAR || algo == PMC_INVERSE_TRIANGULAR_SOLO || algo == PMC_INVERSE_TRIANGULAR_DUO) {
            if (!m->tri_ok) return pmc_fail("pmc_maf_inverse: triangular sweep needs degree groups <= one tile");
            if (m->n_out != 23) return pmc_fail("pmc_maf_inverse: the spline sweeps are built for 8 bins (PMC_INVERSE_NAIVE covers the others)");
            // two wavefronts per 16 rows (D <= 64), else / on request the lone-wave sweep
            if (algo != PMC_INVERSE_TRIANGULAR_SOLO) {
                const int rc = pmc_launch_inverse_nsf2(m, z, x, ladj, n, (hipStream_t)stream);
                if (rc >= 0) return rc;
                if (algo == PMC_INVERSE_TRIANGULAR_DUO) return pmc_fail("pmc_maf_inverse: the two-wave spline sweep needs D <= 64 and its tiles in 160 KiB of LDS");
            }
            return pmc_launch_inverse_tri_nsf(m, z, x, ladj, n, (hipStream_t)stream);
        }
        if (algo == PMC_INVERSE_NAIVE) return pmc_launch_inverse_dpass_wg(m, z, x, ladj, n, (hipStream_t)stream);
        return pmc_fail("pmc_maf_inverse: spline flows know PMC_INVERSE_TRIANGULAR (_SOLO, _DUO) and PMC_INVERSE_NAIVE");
    }
    const int asked = algo;
    if (algo == PMC_INVERSE_AUTO) algo = m->tri_ok ? PMC_INVERSE_TRIANGULAR : PMC_INVERSE_NAIVE;
    if (algo == PMC_INVERSE_TRIANGULAR) {
        if (!m->tri_ok) return pmc_fail("pmc_maf_inverse: triangular sweep needs degree groups <= one tile");
        // register-chain sweeps for output tiles <= 8 (D <= 64), the lane-per-walker sweep for the wider flows
        const int rc = pmc_launch_inverse_tri4(m, z, x, ladj, n, (hipStream_t)stream);
        if (rc >= 0) return rc;
        if (asked != PMC_INVERSE_AUTO) return pmc_fail("pmc_maf_inverse: the triangular sweeps need their tiles in 160 KiB of LDS");
        algo = PMC_INVERSE_NAIVE;                                   // (AUTO: the D-pass algorithm covers what is left)
    } else if (algo == PMC_INVERSE_TRIANGULAR_SOLO || algo == PMC_INVERSE_TRIANGULAR_DUO) {
        if (!m->tri_ok) return pmc_fail("pmc_maf_inverse: triangular sweep needs degree groups <= one tile");
        if (m->nOT > 8) return pmc_fail("pmc_maf_inverse: this sweep needs D <= 64 and its tiles in 160 KiB of LDS");
        const int rc = pmc_launch_inverse_tri4(m, z, x, ladj, n, (hipStream_t)stream, algo == PMC_INVERSE_TRIANGULAR_DUO);
        if (rc >= 0) return rc;
        return pmc_fail("pmc_maf_inverse: this sweep needs D <= 64 and its tiles in 160 KiB of LDS");
    } else if (algo == PMC_INVERSE_TRIANGULAR_LANE || algo == PMC_INVERSE_TRIANGULAR_LANE16) {
        pmc_maf_t mf = *m;
        if (algo == PMC_INVERSE_TRIANGULAR_LANE) mf.lane16 = nullptr;       // (the float32 helpers, whatever is attached)
        else if (!m->lane16 || (m->lane16_fmt != 1 && m->lane16_fmt != 2))
            return pmc_fail("pmc_maf_inverse: PMC_INVERSE_TRIANGULAR_LANE16 needs pmc_maf_t.lane16 (pmc_maf_pack_lane16)");
        const int rc = pmc_launch_tri6(nullptr, &mf, z, x, ladj, n, (hipStream_t)stream);
        if (rc >= 0) return rc;
        return pmc_fail("pmc_maf_inverse: the lane-per-walker sweep needs an affine flow whose degree groups fit a tile");
    }
    if (algo == PMC_INVERSE_NAIVE) {
        const size_t lds = maf_lds_bytes(*m, 3);
        if (int e = set_lds(maf_dense_kernel<1>, lds)) return e;
        hipLaunchKernelGGL(maf_dense_kernel<1>, dim3((unsigned)((n + 15) / 16)), dim3(64), lds,
                           (hipStream_t)stream, *m, z, x, ladj, (float*)nullptr, n);
        return pmc_check_launch("maf_dense_kernel<naive inverse>");
    }
    return pmc_fail("pmc_maf_inverse: unknown algo");
}

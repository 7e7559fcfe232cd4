// Univariate map of the neural spline flows (pocomc/flow.py:69-86 -> zuko.flows.NSF(bins=8)):
// zuko's MonotonicRQSTransform(widths, heights, derivatives, bound=5, slope=1e-3), one thread per
// (feature, row).  The 23 hyper-network outputs of a feature are
//     phi[0:8]  widths,  phi[8:16] heights   -> v / (1 + |2 v / log(slope)|) -> softmax
//                                            -> knots bound * (2 cumsum - 1)
//     phi[16:23] interior derivatives         -> exp(v / (1 + |v / log(slope)|)); end knots: 1
// bin k = searchsorted(knots, x) - 1; outside [0, 8) the map is the identity.  Inside, the
// rational-quadratic spline of Durkan et al. (2019).  Everything is unrolled over the 8 bins with
// compile-time indices so the knot tables stay in registers.
#ifndef PMC_RQS_H
#define PMC_RQS_H

#include <type_traits>
#include "maf_common.h"

#define RQS_K 8
#define RQS_NOUT 23
#define RQS_BOUND 5.0f
// Other bin counts (the reference accepts any zuko flow object, flow.py:87-88): the element code below is templated on the
// number of bins K (3 K - 1 hyper-network outputs per feature); the forward / log_prob kernel, the D-pass inverse and the
// training kernels are instantiated for K = 4, 8, 16, the triangular sweeps for the default K = 8 only.
#define RQS_NOUT_OF(K) (3 * (K) - 1)

struct RqsBin {                 // the selected bin and what the backward pass needs of the tables
    float x0, x1, y0, y1, d0, d1;
    int k;                      // bin index, valid only if inside
    bool inside;
};

template <int K>
struct RqsTablesT {
    float pw[K], ph[K];             // softmax probabilities (bin widths / heights over 2*bound)
    float xk[K + 1], yk[K + 1];
};
typedef RqsTablesT<8> RqsTables;

// Hardware transcendentals (v_rcp_f32 / v_exp_f32 / v_log_f32, ~1 ulp): a lone wave per SIMD pays four
// cycles per VALU instruction, and the IEEE-exact division / expf sequences of ~60 divisions and 18
// exponentials per spline were the bulk of the inverse sweep's time.
#define RQS_INV_LS (1.0f / PMC_LOG_SLOPE)
__device__ __forceinline__ float rqs_rcp(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float rqs_exp(float v) { return __builtin_amdgcn_exp2f(v * 1.4426950408889634f); }
__device__ __forceinline__ float rqs_log(float v) { return __builtin_amdgcn_logf(v) * 0.6931471805599453f; }
__device__ __forceinline__ float rqs_clip2(float v) { return v * rqs_rcp(1.0f + fabsf(v * (2.0f * RQS_INV_LS))); }
__device__ __forceinline__ float rqs_clip1(float v) { return v * rqs_rcp(1.0f + fabsf(v * RQS_INV_LS)); }

template <int K>
__device__ __forceinline__ void rqs_softmax_knots_t(const float* v, float* p, float* knots) {
    float c[K];
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < K; ++j) { c[j] = rqs_clip2(v[j]); mx = fmaxf(mx, c[j]); }
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < K; ++j) { c[j] = rqs_exp(c[j] - mx); sum += c[j]; }
    const float rsum = rqs_rcp(sum);
    float cum = 0.0f;
    knots[0] = -RQS_BOUND;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        p[j] = c[j] * rsum;
        cum += p[j];
        knots[j + 1] = RQS_BOUND * (2.0f * cum - 1.0f);
    }
}

template <int K>
__device__ __forceinline__ void rqs_tables_t(const float* phi, RqsTablesT<K>& t) {
    rqs_softmax_knots_t<K>(phi, t.pw, t.xk);
    rqs_softmax_knots_t<K>(phi + K, t.ph, t.yk);
}

// bin of `v` in `knots` (the x knots for the forward map, the y knots for the inverse)
template <int K>
__device__ __forceinline__ void rqs_select_t(const RqsTablesT<K>& t, const float* phi, const float* knots, float v,
                                             RqsBin& b) {
    b.inside = (v > knots[0]) && (v <= knots[K]);
    b.k = 0;
    b.x0 = t.xk[0]; b.x1 = t.xk[1]; b.y0 = t.yk[0]; b.y1 = t.yk[1];
    float r0 = 0.0f, r1 = rqs_clip1(phi[2 * K]);          // raw (clipped) log-derivatives at the bin's knots
#pragma unroll
    for (int j = 1; j < K; ++j) {
        if (knots[j] < v) {
            b.k = j;
            b.x0 = t.xk[j]; b.x1 = t.xk[j + 1]; b.y0 = t.yk[j]; b.y1 = t.yk[j + 1];
            r0 = rqs_clip1(phi[2 * K + j - 1]);
            r1 = (j + 1 < K) ? rqs_clip1(phi[2 * K + j]) : 0.0f;
        }
    }
    b.d0 = rqs_exp(r0);
    b.d1 = rqs_exp(r1);
}

// y = f(x), ladj = log f'(x)
template <int K>
__device__ __forceinline__ void rqs_forward_t(const float* phi, float x, float& y, float& ladj) {
    RqsTablesT<K> t;
    rqs_tables_t<K>(phi, t);
    RqsBin b;
    rqs_select_t<K>(t, phi, t.xk, x, b);
    const float dx = b.x1 - b.x0, dy = b.y1 - b.y0;
    const float rdx = rqs_rcp(dx);
    const float s = dy * rdx;
    const float z = b.inside ? (x - b.x0) * rdx : 0.0f;
    const float u = z * (1.0f - z);
    const float rden = rqs_rcp(s + (b.d0 + b.d1 - 2.0f * s) * u);
    const float yy = b.y0 + dy * (s * z * z + b.d0 * u) * rden;
    const float jac = s * s * (2.0f * s * u + b.d0 * (1.0f - z) * (1.0f - z) + b.d1 * z * z) * (rden * rden);
    y = b.inside ? yy : x;
    ladj = b.inside ? rqs_log(jac) : 0.0f;
}

// x = f^-1(y), ladj = log f'(x)  (the forward log-derivative at the solution)
template <int K>
__device__ __forceinline__ void rqs_inverse_t(const float* phi, float y, float& x, float& ladj) {
    RqsTablesT<K> t;
    rqs_tables_t<K>(phi, t);
    RqsBin b;
    rqs_select_t<K>(t, phi, t.yk, y, b);
    const float dx = b.x1 - b.x0, dy = b.y1 - b.y0;
    const float s = dy * rqs_rcp(dx);
    const float yr = b.inside ? y - b.y0 : 0.0f;
    const float e = b.d0 + b.d1 - 2.0f * s;
    const float qa = dy * (s - b.d0) + yr * e;
    const float qb = dy * b.d0 - yr * e;
    const float qc = -s * yr;
    const float z = 2.0f * qc * rqs_rcp(-qb - __builtin_amdgcn_sqrtf(qb * qb - 4.0f * qa * qc));
    const float u = z * (1.0f - z);
    const float rden = rqs_rcp(s + e * u);
    const float jac = s * s * (2.0f * s * u + b.d0 * (1.0f - z) * (1.0f - z) + b.d1 * z * z) * (rden * rden);
    x = b.inside ? b.x0 + z * dx : y;
    ladj = b.inside ? rqs_log(jac) : 0.0f;
}

// the reference's default, 8 bins (pocomc/flow.py:71): what the sweep kernels are built for
__device__ __forceinline__ void rqs_softmax_knots(const float* v, float* p, float* knots) { rqs_softmax_knots_t<RQS_K>(v, p, knots); }
__device__ __forceinline__ void rqs_forward(const float* phi, float x, float& y, float& ladj) { rqs_forward_t<RQS_K>(phi, x, y, ladj); }
__device__ __forceinline__ void rqs_inverse(const float* phi, float y, float& x, float& ladj) { rqs_inverse_t<RQS_K>(phi, y, x, ladj); }

// Inverse for the sweep kernel, where the four lanes q = 0..3 of a row would otherwise all repeat the same
// work: lanes with even q build the x-knot table, odd q the y-knot table (the same code on different
// parameters, so the wave does not diverge), publish them in LDS, and every lane then picks the bin and
// the six numbers of it by (run-time) index.  par: the row's 23 parameters in LDS ([0:8] widths, [8:16]
// heights, [16:23] derivatives), tab: 24 floats of LDS scratch of the row.  One wavefront owns the row's
// four lanes, its DS operations execute in order.
__device__ __forceinline__ void rqs_inverse_coop(const float* par, float* tab, int q, float y, float& x, float& ladj) {
    const int side = q & 1;
    float v[RQS_K], pr[RQS_K], kn[RQS_K + 1];
    {
        const float4 a = *reinterpret_cast<const float4*>(par + 8 * side), b = *reinterpret_cast<const float4*>(par + 8 * side + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    rqs_softmax_knots(v, pr, kn);
    if (q < 2) {
        float* t = tab + 12 * side;
        *reinterpret_cast<float4*>(t) = make_float4(kn[0], kn[1], kn[2], kn[3]);
        *reinterpret_cast<float4*>(t + 4) = make_float4(kn[4], kn[5], kn[6], kn[7]);
        t[8] = kn[8];
    }
    WAVE_LDS_FENCE();
    float yk[RQS_K + 1];
    {
        const float4 a = *reinterpret_cast<const float4*>(tab + 12), b = *reinterpret_cast<const float4*>(tab + 16);
        yk[0] = a.x; yk[1] = a.y; yk[2] = a.z; yk[3] = a.w; yk[4] = b.x; yk[5] = b.y; yk[6] = b.z; yk[7] = b.w;
        yk[8] = tab[20];
    }
    const bool inside = (y > yk[0]) && (y <= yk[RQS_K]);
    int k = 0;
    float y0 = yk[0], y1 = yk[1];
#pragma unroll
    for (int j = 1; j < RQS_K; ++j)
        if (yk[j] < y) { k = j; y0 = yk[j]; y1 = yk[j + 1]; }
    const float x0 = tab[k], x1 = tab[k + 1];
    const float r0 = (k >= 1) ? rqs_clip1(par[2 * RQS_K + k - 1]) : 0.0f;
    const float r1 = (k + 1 < RQS_K) ? rqs_clip1(par[2 * RQS_K + (k + 1 < RQS_K ? k : 0)]) : 0.0f;
    const float d0 = rqs_exp(r0), d1 = rqs_exp(r1);
    const float dx = x1 - x0, dy = y1 - y0;
    const float s = dy * rqs_rcp(dx);
    const float yr = inside ? y - y0 : 0.0f;
    const float e = d0 + d1 - 2.0f * s;
    const float qa = dy * (s - d0) + yr * e;
    const float qb = dy * d0 - yr * e;
    const float qc = -s * yr;
    const float z = 2.0f * qc * rqs_rcp(-qb - __builtin_amdgcn_sqrtf(qb * qb - 4.0f * qa * qc));
    const float u = z * (1.0f - z);
    const float rden = rqs_rcp(s + e * u);
    const float jac = s * s * (2.0f * s * u + d0 * (1.0f - z) * (1.0f - z) + d1 * z * z) * (rden * rden);
    x = inside ? x0 + z * dx : y;
    ladj = inside ? rqs_log(jac) : 0.0f;
}

// Inverse for the two-wave sweep (maf_inverse_nsf2.hip), where lane (q, p) of the chain wave holds rows 4q .. 4q+3 of the
// rank's two output tiles for walker p (o0: widths 0-3 | widths 4-7 | heights 0-3 | heights 4-7 by q; o1: derivatives
// 0-3 | 4-6 | - | -): every lane soft-clips and exponentiates ITS four widths / heights before the exchange (the soft clip
// bounds them by |log slope| / 2 = 3.45, so the softmax needs no maximum), the panel row carries the sixteen unnormalised
// weights and the raw derivatives, and behind the ONE exchange every lane finds the bin on unnormalised cumulative sums
// (knot_j < y  <=>  cum_j < (y + B) / 2B * sum) and normalises only the four knots of that bin.  par: 32 floats of LDS.
//
// Round 4: the solve is written in STAGES with fourteen numbered slots between them -- `sh(integral_constant<K>)` is the
// caller's work for the shadow of stage K (the chain wave issues the NEXT rank's MFMAs there: one per slot, each behind
// >= 32 cycles of vector work or inside an LDS wait, so the matrix pipe never stalls the solve) -- and the
// log-derivative, which nothing downstream waits for, is returned UNEVALUATED (RqsPend -> rqs_ladj_1 / _2 / _3, run by the
// caller in the shadows of the next rank's hops).  Scheduling fences keep the stages in this order; inside a stage the
// four (or two) independent lines are written side by side so that dependent instructions stand four apart.
struct RqsPend { float s, e, z, d0, d1, u, rden, jac; bool inside; };
__device__ __forceinline__ void rqs_ladj_1(RqsPend& r) { r.u = r.z * (1.0f - r.z); r.rden = rqs_rcp(r.s + r.e * r.u); }
__device__ __forceinline__ void rqs_ladj_2(RqsPend& r) {
    r.jac = r.s * r.s * (2.0f * r.s * r.u + r.d0 * (1.0f - r.z) * (1.0f - r.z) + r.d1 * r.z * r.z) * (r.rden * r.rden);
}
__device__ __forceinline__ float rqs_ladj_3(const RqsPend& r) { return r.inside ? rqs_log(r.jac) : 0.0f; }

struct RqsNoShadow {
    template <int K> __device__ __forceinline__ void operator()(std::integral_constant<int, K>) const {}
};
#define RQS_NSLOTS 14
#define RQS_SLOT(K) { __builtin_amdgcn_sched_barrier(0); sh(std::integral_constant<int, K>{}); __builtin_amdgcn_sched_barrier(0); }

template <class SH>
__device__ __forceinline__ void rqs_inverse_split_sh(const f32x4& o0, const f32x4& o1, float* par, int q, float y, float& x, RqsPend& pend,
                                                     const SH& sh) {
    // ---- A: soft clip and exponential of the lane's four widths / heights (rqs_exp(rqs_clip2(v)), four lines side by side)
    const float t0 = o0[0] * (2.0f * RQS_INV_LS), t1 = o0[1] * (2.0f * RQS_INV_LS), t2 = o0[2] * (2.0f * RQS_INV_LS), t3 = o0[3] * (2.0f * RQS_INV_LS);
    RQS_SLOT(0)
    const float r0 = rqs_rcp(1.0f + fabsf(t0)), r1 = rqs_rcp(1.0f + fabsf(t1));
    RQS_SLOT(1)
    const float r2 = rqs_rcp(1.0f + fabsf(t2)), r3 = rqs_rcp(1.0f + fabsf(t3));
    RQS_SLOT(2)
    const float m0 = (o0[0] * r0) * 1.4426950408889634f, m1 = (o0[1] * r1) * 1.4426950408889634f;
    const float m2 = (o0[2] * r2) * 1.4426950408889634f, m3 = (o0[3] * r3) * 1.4426950408889634f;
    RQS_SLOT(3)
    float4 ex;
    ex.x = __builtin_amdgcn_exp2f(m0); ex.y = __builtin_amdgcn_exp2f(m1);
    RQS_SLOT(4)
    ex.z = __builtin_amdgcn_exp2f(m2); ex.w = __builtin_amdgcn_exp2f(m3);
    // ---- B: the exchange
    *reinterpret_cast<float4*>(par + 4 * q) = ex;
    *reinterpret_cast<float4*>(par + 16 + 4 * q) = make_float4(o1[0], o1[1], o1[2], o1[3]);
    WAVE_LDS_FENCE();
    const float4 h0 = *reinterpret_cast<const float4*>(par + 8), h1 = *reinterpret_cast<const float4*>(par + 12);
    const float4 w0 = *reinterpret_cast<const float4*>(par), w1 = *reinterpret_cast<const float4*>(par + 4);
    const bool inside = (y > -RQS_BOUND) && (y <= RQS_BOUND);
    const float ty = (y + RQS_BOUND) * (0.5f / RQS_BOUND);
    RQS_SLOT(5)
    RQS_SLOT(6)
    RQS_SLOT(7)
    // ---- C: unnormalised cumulative sums, heights and widths side by side
    const float ch1 = h0.x, cw1 = w0.x;
    const float ch2 = ch1 + h0.y, cw2 = cw1 + w0.y;
    const float ch3 = ch2 + h0.z, cw3 = cw2 + w0.z;
    const float ch4 = ch3 + h0.w, cw4 = cw3 + w0.w;
    RQS_SLOT(8)
    const float ch5 = ch4 + h1.x, cw5 = cw4 + w1.x;
    const float ch6 = ch5 + h1.y, cw6 = cw5 + w1.y;
    const float ch7 = ch6 + h1.z, cw7 = cw6 + w1.z;
    const float hs = ch7 + h1.w, ws = cw7 + w1.w;
    RQS_SLOT(9)
    const float t = ty * hs;
    const float rh = rqs_rcp(hs) * (2.0f * RQS_BOUND), rw = rqs_rcp(ws) * (2.0f * RQS_BOUND);
    // ---- D: bisection over the nine knots (three compares, 20 selects), then the bin's two raw derivatives by an
    // indexed LDS read (selecting them from registers next to the knots cost more: a lone wavefront pays ~10 cycles per
    // dependent select)
    const bool c4 = ch4 < t;
    const float e0 = c4 ? ch4 : 0.0f, e1 = c4 ? ch5 : ch1, e2 = c4 ? ch6 : ch2, e3 = c4 ? ch7 : ch3, e4 = c4 ? hs : ch4;
    const float f0 = c4 ? cw4 : 0.0f, f1 = c4 ? cw5 : cw1, f2 = c4 ? cw6 : cw2, f3 = c4 ? cw7 : cw3, f4 = c4 ? ws : cw4;
    RQS_SLOT(10)
    const bool c2 = e2 < t;
    const float g0 = c2 ? e2 : e0, g1 = c2 ? e3 : e1, g2 = c2 ? e4 : e2;
    const float i0 = c2 ? f2 : f0, i1 = c2 ? f3 : f1, i2 = c2 ? f4 : f2;
    const bool c1 = g1 < t;
    const float a0 = c1 ? g1 : g0, a1 = c1 ? g2 : g1, b0 = c1 ? i1 : i0, b1 = c1 ? i2 : i1;
    const int k = (c4 ? 4 : 0) + (c2 ? 2 : 0) + (c1 ? 1 : 0);
    const float q0 = par[2 * RQS_K + (k >= 1 ? k - 1 : 0)], q1 = par[2 * RQS_K + (k + 1 < RQS_K ? k : 0)];
    // the bin's own unnormalised width and height (two more indexed reads of the panel row): dx and dy as w_k / sum and
    // h_k / sum carry the rounding of ONE weight -- as differences of knots near +-5 they carried ~eps * 5 each, 1e-5 of a
    // narrow bin (round 5: the step's worst walker against the float64 evaluation, 1.13e-5 -> see tests/test_gpu_mcmc.py)
    const float wk = par[k], hk = par[RQS_K + k];
    (void)a1; (void)b1;
    RQS_SLOT(11)
    // ---- E: the bin's knots (while the derivatives are on their way)
    const float y0 = a0 * rh - RQS_BOUND, x0 = b0 * rw - RQS_BOUND;
    const float dx = wk * rw, dy = hk * rh;
    const float s = dy * rqs_rcp(dx);
    const float yr = inside ? y - y0 : 0.0f;
    RQS_SLOT(12)
    // ---- F: derivatives at the bin's knots; the end knots' raw log-derivative is 0 (derivative 1)
    const float v0 = (k >= 1) ? rqs_clip1(q0) : 0.0f, v1 = (k + 1 < RQS_K) ? rqs_clip1(q1) : 0.0f;
    const float d0 = rqs_exp(v0), d1 = rqs_exp(v1);
    RQS_SLOT(13)
    // ---- G: Durkan's inverse
    const float e = d0 + d1 - 2.0f * s;
    const float qa = dy * (s - d0) + yr * e;
    const float qb = dy * d0 - yr * e;
    const float qc = -s * yr;
    const float z = 2.0f * qc * rqs_rcp(-qb - __builtin_amdgcn_sqrtf(qb * qb - 4.0f * qa * qc));
    x = inside ? x0 + z * dx : y;
    pend.s = s; pend.e = e; pend.z = z; pend.d0 = d0; pend.d1 = d1; pend.inside = inside;
}

__device__ __forceinline__ void rqs_inverse_split(const f32x4& o0, const f32x4& o1, float* par, int q, float y, float& x, float& ladj) {
    RqsPend r;
    rqs_inverse_split_sh(o0, o1, par, q, y, x, r, RqsNoShadow{});
    rqs_ladj_1(r); rqs_ladj_2(r);
    ladj = rqs_ladj_3(r);
}

// Reverse mode of F = gy * y + gl * ladj:  dphi[3 K - 1] and gx = dF/dx.
template <int K>
__device__ __forceinline__ void rqs_backward_t(const float* phi, float x, float gy, float gl, float* dphi, float& gx) {
    RqsTablesT<K> t;
    rqs_tables_t<K>(phi, t);
    RqsBin b;
    rqs_select_t<K>(t, phi, t.xk, x, b);
#pragma unroll
    for (int j = 0; j < (3 * K - 1); ++j) dphi[j] = 0.0f;
    gx = gy;
    if (!b.inside) return;
    const float dx = b.x1 - b.x0, dy = b.y1 - b.y0;
    const float rdx = rqs_rcp(dx);
    const float s = dy * rdx;
    const float z = (x - b.x0) * rdx;
    const float u = z * (1.0f - z);
    const float e = b.d0 + b.d1 - 2.0f * s;
    const float num = s * z * z + b.d0 * u;
    const float rden = rqs_rcp(s + e * u);
    const float jn = 2.0f * s * u + b.d0 * (1.0f - z) * (1.0f - z) + b.d1 * z * z;
    // adjoints of the spline formula
    const float g_num = gy * dy * rden;
    const float g_den = -gy * dy * num * (rden * rden) - 2.0f * gl * rden;
    const float g_jn = gl * rqs_rcp(jn);
    float g_dy = gy * num * rden;
    const float g_s = g_num * z * z + g_den * (1.0f - 2.0f * u) + 2.0f * gl * rqs_rcp(s) + g_jn * 2.0f * u;
    const float g_d0 = (g_num + g_den) * u + g_jn * (1.0f - z) * (1.0f - z);
    const float g_d1 = g_den * u + g_jn * z * z;
    const float g_u = g_num * b.d0 + g_den * e + g_jn * 2.0f * s;
    const float g_z = g_num * 2.0f * s * z + g_u * (1.0f - 2.0f * z) + g_jn * (-2.0f * b.d0 * (1.0f - z) + 2.0f * b.d1 * z);
    // s = dy / dx,  z = (x - x0) / dx
    g_dy += g_s * rdx;
    const float g_dx = -(g_s * s + g_z * z) * rdx;
    gx = g_z * rdx;
    const float g_x0 = -g_z * rdx - g_dx, g_x1 = g_dx;
    const float g_y0 = gy - g_dy, g_y1 = g_dy;
    // knots -> softmax probabilities: knot_j = B (2 sum_{i<j} p_i - 1)
    //   dF/dp_i = 2B (g_k0 [i < k] + g_k1 [i < k+1])
    float gpw[K], gph[K];
    float dotw = 0.0f, doth = 0.0f;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const float lo = (i < b.k) ? 1.0f : 0.0f, hi = (i <= b.k) ? 1.0f : 0.0f;
        gpw[i] = 2.0f * RQS_BOUND * (g_x0 * lo + g_x1 * hi);
        gph[i] = 2.0f * RQS_BOUND * (g_y0 * lo + g_y1 * hi);
        dotw += t.pw[i] * gpw[i];
        doth += t.ph[i] * gph[i];
    }
#pragma unroll
    for (int i = 0; i < K; ++i) {
        // softmax backward, then the soft clip v / (1 + |2v/ls|) whose derivative is 1 / (1 + |2v/ls|)^2
        const float cw = rqs_rcp(1.0f + fabsf(phi[i] * (2.0f * RQS_INV_LS)));
        const float ch = rqs_rcp(1.0f + fabsf(phi[K + i] * (2.0f * RQS_INV_LS)));
        dphi[i] = t.pw[i] * (gpw[i] - dotw) * (cw * cw);
        dphi[K + i] = t.ph[i] * (gph[i] - doth) * (ch * ch);
    }
    // derivatives: d = exp(clip1(v)); knot k uses phi[2 K + k - 1] (k >= 1), knot k+1 uses phi[2 K + k] (k + 1 <= K - 1)
#pragma unroll
    for (int j = 0; j < K - 1; ++j) {
        const float c = rqs_rcp(1.0f + fabsf(phi[2 * K + j] * RQS_INV_LS));
        float g = 0.0f;
        if (j == b.k - 1) g = g_d0 * b.d0;
        if (j == b.k) g = g_d1 * b.d1;
        dphi[2 * K + j] = g * (c * c);
    }
}

__device__ __forceinline__ void rqs_backward(const float* phi, float x, float gy, float gl, float* dphi, float& gx) {
    rqs_backward_t<RQS_K>(phi, x, gy, gl, dphi, gx);
}

#endif

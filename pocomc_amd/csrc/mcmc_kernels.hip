// Per-particle kernels of the MCMC step (pocomc/mcmc.py) and the scaler inside it
// (pocomc/scaler.py).  float64 like the reference's numpy arithmetic; the
// expressions that decide accept/reject keep the reference's operation order and
// are compiled without FMA contraction so that a replayed step makes the same
// decisions as the reference.
//
// These are HBM/L2-streaming sweeps with wave-level reductions -- no MFMA here.

#include <hip/hip_runtime.h>
#include <string.h>
#include <stdio.h>
#include <fcntl.h>
#include <unistd.h>
#include <time.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <math.h>
#include <stdint.h>
#include "philox.h"
#include "pmc_internal.h"

// The reference evaluates these expressions with numpy: one rounding per operation.
#pragma clang fp contract(off)

#define PROP_ROWS 64          // particles per block in propose_kernel (one wave)
#define ACC_ROWS 64           // particles per block in accept_kernel
#ifndef SCL_ROWS
#define SCL_ROWS 64           // particles per block in the scaler kernels
#endif

// ===========================================================================
// proposal: mcmc.py:77-85 (tpCN), :251-253 / :561-563 (RWM)
// ===========================================================================
__global__ __launch_bounds__(PROP_ROWS) void propose_kernel(
    int kind, const float* __restrict__ cur32, const double* __restrict__ cur64,
    const double* __restrict__ mu, const double* __restrict__ inv_cov, const double* __restrict__ chol,
    double nu, double sigma, double cn_a, pmc_rng_t rng, double* __restrict__ prop64,
    float* __restrict__ prop32, double* __restrict__ quad, double* __restrict__ quad_prop,
    int64_t n, int D, const double* __restrict__ adapt) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (adapt) { sigma = adapt[0]; cn_a = adapt[1]; mu = adapt + 2; }   // pmc_step_t.adapt_state
    const int LD = PROP_ROWS + 1;
    double* dif = sm;                 // [D][LD]  theta - mu (tpCN) or theta (RWM)
    double* zz = sm + (size_t)D * LD; // [D][LD]  z, overwritten by the proposal
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * PROP_ROWS;
    const int rows = (int)min((int64_t)PROP_ROWS, n - row0);

    // coalesced load of the block's [rows][D] slab
    for (int e = tid; e < rows * D; e += PROP_ROWS) {
        const int r = e / D, j = e - r * D;
        const double v = cur32 ? (double)cur32[(row0 + r) * D + j] : cur64[(row0 + r) * D + j];
        dif[j * LD + r] = (kind == PMC_KIND_TPCN) ? v - mu[j] : v;
        if (rng.normal) zz[j * LD + r] = rng.normal[(row0 + r) * D + j];
    }
    __syncthreads();
    if (tid >= rows) return;
    const int64_t gidx = rng.offset + row0 + tid;
    if (!rng.normal) {
        Philox ph(rng.seed, rng.step, gidx, 1);
        for (int j = 0; j < D; j += 2) {
            double a, b;
            ph.normal2(a, b);
            zz[j * LD + tid] = a;
            if (j + 1 < D) zz[(j + 1) * LD + tid] = b;
        }
    }

    double scale_z;                   // multiplies L z
    double q_cur = 0.0;
    if (kind == PMC_KIND_TPCN) {
        // diff^T inv_cov diff  (mcmc.py:80)
        for (int i = 0; i < D; ++i) {
            double w = 0.0;
            const double* Si = inv_cov + (size_t)i * D;
            for (int j = 0; j < D; ++j) w += Si[j] * dif[j * LD + tid];
            q_cur += dif[i * LD + tid] * w;
        }
        double g;
        if (rng.gamma) g = rng.gamma[row0 + tid];
        else { Philox ph(rng.seed, rng.step, gidx, 0); g = ph.std_gamma(0.5 * ((double)D + nu)); }
        const double s = 1.0 / ((2.0 / (nu + q_cur)) * g);         // 1/np.random.gamma(shape, scale)
        scale_z = sigma * sqrt(s);
    } else {
        scale_z = sigma;
    }

    // proposal, last coordinate first so that z can be overwritten in place
    for (int i = D - 1; i >= 0; --i) {
        double lz = 0.0;
        const double* Li = chol + (size_t)i * D;
        for (int j = 0; j <= i; ++j) lz += Li[j] * zz[j * LD + tid];
        double v;
        if (kind == PMC_KIND_TPCN) v = (mu[i] + cn_a * dif[i * LD + tid]) + scale_z * lz;
        else v = dif[i * LD + tid] + scale_z * lz;
        zz[i * LD + tid] = v;
    }
    if (kind == PMC_KIND_TPCN) {
        double q_new = 0.0;
        for (int i = 0; i < D; ++i) {
            double w = 0.0;
            const double* Si = inv_cov + (size_t)i * D;
            for (int j = 0; j < D; ++j) w += Si[j] * (zz[j * LD + tid] - mu[j]);
            q_new += (zz[i * LD + tid] - mu[i]) * w;
        }
        if (quad) quad[row0 + tid] = q_cur;
        if (quad_prop) quad_prop[row0 + tid] = q_new;
    }
    // each lane writes its own row: D*8 B per lane, whole cache lines per lane
    for (int j = 0; j < D; ++j) {
        const double v = zz[j * LD + tid];
        if (prop64) prop64[(row0 + tid) * D + j] = v;
        if (prop32) prop32[(row0 + tid) * D + j] = (float)v;
    }
}

#include "scaler_body.h"

// SCL_THREADS threads per block: the probit / logit maps are long dependent f64 chains, so the block's 64 x D elements
// are spread over as many lanes as a block can have (four elements per lane at D = 32 instead of eight; 1024 threads would halve the registers of a lane and spill)
#define SCL_THREADS 512
__global__ __launch_bounds__(SCL_THREADS) void scaler_inverse_kernel(
    pmc_scaler_t s, const float* __restrict__ u_in, const double* __restrict__ u_in64,
    double* __restrict__ u_out, double* __restrict__ x_out, double* __restrict__ x_colmajor,
    double* __restrict__ ldj_out, int32_t* __restrict__ finite_out, int64_t n, pmc_prior_t pr,
    double* __restrict__ logp_out, int32_t* __restrict__ finite_copy, double* __restrict__ logp_copy,
    unsigned* __restrict__ done_ticket, long long* __restrict__ done_flag, long long done_value, pmc_scaler_extra ex) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int D = s.D;
    const bool keep_x = x_colmajor || logp_out;   // x' stays in LDS for the column-major copy / the fused prior
    double* Jt = sm;                              // [SCL_ROWS][D]
    double* Xt = sm + (size_t)SCL_ROWS * D;       // [D][SCL_ROWS+1]  (only when keep_x)
    int* rowfin = reinterpret_cast<int*>(sm + (size_t)SCL_ROWS * D + (keep_x ? (size_t)D * (SCL_ROWS + 1) : 0));
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * SCL_ROWS;
    const int rows = (int)min((int64_t)SCL_ROWS, n - row0);
    if (tid < SCL_ROWS) rowfin[tid] = 1;
    __syncthreads();
    for (int e = tid; e < rows * D; e += SCL_THREADS) {
        const int r = e / D, j = e - r * D;
        const int64_t g = (row0 + r) * D + j;
        double u = u_in ? (double)u_in[g] : u_in64[g];
        double t, x, J;
        t = s.scale ? s.mu[j] + s.sigma[j] * u : u;
        bound_inverse(s, j, t, x, J);
        if (s.bc) {
            // mcmc.py:94-97: wrap x, re-derive u from it, invert again
            x = apply_bc(s, j, x);
            u = bound_forward(s, j, x);
            t = s.scale ? s.mu[j] + s.sigma[j] * u : u;
            bound_inverse(s, j, t, x, J);
        }
        u_out[g] = u;
        x_out[g] = x;
        if (keep_x) Xt[j * (SCL_ROWS + 1) + r] = x;
        Jt[r * D + j] = J;
        if (!isfinite(x)) rowfin[r] = 0;
    }
    __syncthreads();
    if (tid < rows) {
        double l = np_pairwise_sum(Jt + (size_t)tid * D, D);
        if (s.scale) l = s.sum_log_sigma + l;
        ldj_out[row0 + tid] = l;
        const int fin = (rowfin[tid] && isfinite(l)) ? 1 : 0;
        finite_out[row0 + tid] = fin;
        if (finite_copy) finite_copy[row0 + tid] = fin;
        if (logp_out) {
            // Prior.logpdf of the finite rows (mcmc.py:105-107), dimension after dimension like prior_logpdf_kernel
            double lp = -INFINITY;
            if (fin) {
                lp = 0.0;
                for (int j = 0; j < D; ++j) lp += prior_term(pr, j, Xt[j * (SCL_ROWS + 1) + tid]);
            }
            logp_out[row0 + tid] = lp;
            if (logp_copy) logp_copy[row0 + tid] = lp;
            if (!isfinite(lp)) rowfin[tid] = 0;
        }
        // rowfin: from here on "the row reaches the likelihood" (mcmc.py:100-109's two masks)
        if (!fin) rowfin[tid] = 0;
        if (ex.bad_count && !rowfin[tid]) atomicAdd(ex.bad_count, 1u);
    }
    if (x_colmajor) {
        if (ex.fill_x) __syncthreads();
        // host copy of x' in column-major order ((n, D) Fortran array on the host): coalesced along rows.  A row that does
        // not reach the likelihood carries the walker's CURRENT x there when asked (pmc_step_t.fill_rejected): the host can
        // hand the whole block to the likelihood and drop those rows' values instead of gathering the others
        for (int e = tid; e < rows * D; e += SCL_THREADS) {
            const int j = e / rows, r = e - j * rows;
            x_colmajor[(size_t)j * n + row0 + r] = (ex.fill_x && !rowfin[r]) ? ex.fill_x[(row0 + r) * D + j] : Xt[j * (SCL_ROWS + 1) + r];
        }
    }
    if (done_flag) {
        // completion word in pinned host memory: every thread waits for the acknowledgement of its stores (agent-scope
        // release; see scaler_body.h), the block draws a ticket, the last one publishes done_value with a system-scope
        // release (the host spins on it instead of going through the runtime)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) {
            const unsigned t = __hip_atomic_fetch_add(done_ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (t == gridDim.x - 1) {
                *done_ticket = 0u;
                if (ex.bad_count && ex.bad_flag) {        // (every block's count is in: its ticket came behind its atomicAdd)
                    const unsigned bad = __hip_atomic_exchange(ex.bad_count, 0u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                    *ex.bad_flag = (long long)bad;
                }
                __threadfence_system();
                __hip_atomic_store(done_flag, done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

__global__ __launch_bounds__(256) void scaler_forward_kernel(pmc_scaler_t s, const double* __restrict__ x,
                                                             double* __restrict__ u, int64_t n) {
    const int D = s.D;
    const int64_t total = n * D;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int j = (int)(e % D);
        u[e] = bound_forward(s, j, x[e]);
    }
}

// ===========================================================================
// Prior.logpdf (pocomc/prior.py:70-100) for the scipy.stats families the device knows:
// logp = sum_j dist_j.logpdf(x[:, j]), accumulated dimension after dimension like the reference
// ===========================================================================
__global__ __launch_bounds__(256) void prior_logpdf_kernel(pmc_prior_t pr, const double* __restrict__ x,
                                                           const int32_t* __restrict__ finite,
                                                           double* __restrict__ logp, int64_t n) {
    const int D = pr.D;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        double lp = -INFINITY;
        if (!finite || finite[k]) {                                  // mcmc.py:105-107
            lp = 0.0;
            const double* xr = x + k * D;
            for (int j = 0; j < D; ++j) lp += prior_term(pr, j, xr[j]);
        }
        logp[k] = lp;
    }
}

extern "C" int pmc_prior_logpdf(const pmc_prior_t* pr, const double* x, const int32_t* finite, double* logp,
                                int64_t n, void* stream) {
    if (!pr || !pr->family || !pr->loc || !pr->scale || pr->D < 1 || !x || !logp || n < 0)
        return pmc_fail("pmc_prior_logpdf: bad argument");
    if (n == 0) return 0;
    int64_t grid = (n + 255) / 256; if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(prior_logpdf_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, *pr, x, finite,
                       logp, n);
    return pmc_check_launch("prior_logpdf_kernel");
}

// ===========================================================================
// Metropolis ratio, accept, reductions: mcmc.py:124-156 and variants
// ===========================================================================
__device__ __forceinline__ double block_sum_256(double v, double* red, int tid) {
    // deterministic tree: wave shuffle, then 4 wave totals in fixed order
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Block = 256 threads = ACC_ROWS (64) walkers.  Wave 0 evaluates the Metropolis ratio of the block's
// walkers; then all four waves copy the accepted rows (independent, unrolled loads) and take the
// column sums on the fly.  The last block to arrive (agent-scope release -> ticket -> acquire) folds
// the per-block partials in a fixed order, so the result is deterministic and needs no second launch.
__device__ __forceinline__ void adapt_apply(const pmc_adapt_args& ad, const double* sums, int tid, int D) {
    // numpy's expressions of mcmc.py:152-156 in their operation order, no contraction
#pragma clang fp contract(off)
    if (tid == 0) {
        const double mean_alpha = sums[0] / ad.n_total;
        const double sg = ad.state[0];
        double sn = sg + ad.c_sigma * (mean_alpha - 0.234);
        const int how = ad.mode & 7;
        if (how == PMC_ADAPT_TPCN) sn = fabs(fmin(sn, ad.cap));
        else if (how == PMC_ADAPT_RWM) sn = fabs(sn);
        ad.state[0] = sn;
        ad.state[1] = (how == PMC_ADAPT_TPCN) ? sqrt(1.0 - sn * sn) : 0.0;          // (1 - sigma^2)^0.5, mcmc.py:85
    }
    if ((ad.mode & PMC_ADAPT_MU) && tid < D) {
        const float mean_theta = (float)(sums[4 + tid] / ad.n_total);              // np.mean of the float32 theta
        const double m = ad.state[2 + tid];
        ad.state[2 + tid] = m + ad.c_mu * ((double)mean_theta - m);
    }
}

__global__ __launch_bounds__(256) void accept_kernel(
    int preconditioned, int tpcn, pmc_state_t cur, pmc_proposal_t prop, double beta, double nu,
    pmc_rng_t rng, double* __restrict__ alpha_out, int32_t* __restrict__ accept_out,
    double* __restrict__ partials, unsigned* __restrict__ ticket, double* __restrict__ sums,
    double* __restrict__ sums_copy, long long* __restrict__ done_flag, long long done_value, int64_t n, int D,
    pmc_adapt_args ad) {
    __shared__ int flag[ACC_ROWS];
    __shared__ double colsum[8][33];
    __shared__ int is_last;
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * ACC_ROWS;
    const int rows = (int)min((int64_t)ACC_ROWS, n - row0);
    const int W = D + 4;
    double* P = partials + (size_t)blockIdx.x * W;

    if (tid < 64) {
        const int64_t k = row0 + tid;
        double alpha = 0.0, lp_post = 0.0, ldj_post = 0.0;
        int acc = 0;
        if (tid < rows) {
            const double logl = cur.logl[k], logp = cur.logp[k], ldj = cur.logdetj[k];
            const double logl_p = prop.logl[k], logp_p = prop.logp[k], ldj_p = prop.logdetj[k];
            double e;
            {
                // mcmc.py:130-133, left to right
                e = logl_p * beta - logl * beta + logp_p - logp + ldj_p - ldj;
                if (preconditioned) e = e + (double)prop.logdetj_flow[k] - (double)cur.logdetj_flow[k];
                if (tpcn) {
                    const double c = -((double)D + nu) / 2.0;
                    const double A = c * log(1.0 + prop.quad_prop[k] / nu);     // mcmc.py:128
                    const double B = c * log(1.0 + prop.quad[k] / nu);          // mcmc.py:129
                    e = e - A + B;
                }
            }
            const double v = exp(e);
            alpha = (v != v) ? 0.0 : (v < 1.0 ? v : 1.0);      // np.minimum(1, .); NaN -> 0 (mcmc.py:134)
            double ur;
            if (rng.uniform) ur = rng.uniform[k];
            else { Philox ph(rng.seed, rng.step, rng.offset + k, 2); double d; ph.uniform2(ur, d); }
            acc = ur < alpha;
            if (acc) {
                cur.logdetj[k] = ldj_p; cur.logl[k] = logl_p; cur.logp[k] = logp_p;
                if (preconditioned) cur.logdetj_flow[k] = prop.logdetj_flow[k];
            }
            lp_post = acc ? (logl_p + logp_p) : (logl + logp);
            ldj_post = lp_post + (acc ? ldj_p : ldj);             // logl + logp + logdetj (mcmc.py:243, :327)
            if (alpha_out) alpha_out[k] = alpha;
            if (accept_out) accept_out[k] = acc;
        }
        flag[tid] = acc;
        double s0 = alpha, s1 = lp_post, s2 = ldj_post, s3 = (double)acc;
        for (int o = 32; o > 0; o >>= 1) {
            s0 += __shfl_down(s0, o); s1 += __shfl_down(s1, o); s2 += __shfl_down(s2, o); s3 += __shfl_down(s3, o);
        }
        if (tid == 0) { P[0] = s0; P[1] = s1; P[2] = s2; P[3] = s3; }
    }
    __syncthreads();

    // accepted rows overwrite theta / u / x; column sums of the moved variable on the fly
    const int tx = tid & 31, ty = tid >> 5;
    for (int j0 = 0; j0 < D; j0 += 32) {
        const int j = j0 + tx;
        double csum = 0.0;
        if (j < D) {
            // all eight rows' operands are requested before the first is used (one row at a time each load pair waited for
            // the previous row's: eight L2 round trips in a row); the sums keep their order
            double pu[ACC_ROWS / 8], px[ACC_ROWS / 8], cu[ACC_ROWS / 8];
            float thn[ACC_ROWS / 8], tho[ACC_ROWS / 8];
#pragma unroll
            for (int i = 0; i < ACC_ROWS / 8; ++i) {
                const int r = ty + 8 * i;
                pu[i] = px[i] = cu[i] = 0.0; thn[i] = tho[i] = 0.0f;
                if (r < rows) {
                    const int64_t g = (row0 + r) * D + j;
                    pu[i] = prop.u[g]; px[i] = prop.x[g];
                    if (preconditioned) { thn[i] = (float)prop.theta64[g]; tho[i] = cur.theta32[g]; }   // theta is a float32 array (tools.py:339)
                    else cu[i] = cur.u[g];
                }
            }
#pragma unroll
            for (int i = 0; i < ACC_ROWS / 8; ++i) {
                const int r = ty + 8 * i;
                if (r < rows) {
                    const int64_t g = (row0 + r) * D + j;
                    const int a = flag[r];
                    if (a) { cur.u[g] = pu[i]; cur.x[g] = px[i]; }
                    if (preconditioned) {
                        if (a) cur.theta32[g] = thn[i];
                        csum += (double)(a ? thn[i] : tho[i]);
                    } else {
                        csum += a ? pu[i] : cu[i];
                    }
                }
            }
        }
        colsum[ty][tx] = csum;
        __syncthreads();
        if (ty == 0 && j < D) {
            double t = 0.0;
            for (int y = 0; y < 8; ++y) t += colsum[y][tx];
            P[4 + j] = t;
        }
        __syncthreads();
    }

    // ---- publish this block's partials, draw a ticket; the last arriver reduces (guide G16)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (t == gridDim.x - 1);
        if (is_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!is_last) return;
    const int nb = gridDim.x;
    __shared__ double fold[256];
    for (int c0 = 0; c0 < W; c0 += 256) {
        // S threads per column, each folding every S-th block; then the S partial sums in order
        const int Wc = min(256, W - c0);
        const int S = max(1, 256 / Wc);
        const int c = tid % Wc, sidx = tid / Wc;
        double t = 0.0;
        if (sidx < S) {
            // (the partials of 16 blocks requested at once, added in the same order as one by one)
            for (int b0 = sidx; b0 < nb; b0 += 16 * S) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int b2 = b0 + u * S;
                    v[u] = b2 < nb ? partials[(size_t)b2 * W + c0 + c] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (b0 + u * S < nb) t += v[u];
            }
        }
        fold[tid] = (sidx < S) ? t : 0.0;
        __syncthreads();
        if (tid < Wc) {
            double tot = 0.0;
            for (int s2 = 0; s2 < S; ++s2) tot += fold[s2 * Wc + tid];
            sums[c0 + tid] = tot;
            if (sums_copy) sums_copy[c0 + tid] = tot;
        }
        __syncthreads();
    }
    if (ad.n_other > 0) {
        // this launch closes a walker set stepped as row ranges: total = these sums + the other ranges' (their
        // accept kernels precede this one on the stream); the host copy and the adaptation see the total
        __shared__ double total[260];
        for (int j = tid; j < W; j += 256) {
            double t = sums[j];
            for (int k = 0; k < ad.n_other; ++k) t += ad.other[k][j];
            total[j] = t;
            if (sums_copy) sums_copy[j] = t;
        }
        __syncthreads();
        if (ad.state && ad.mode) adapt_apply(ad, total, tid, D);
    } else if (ad.state && ad.mode) adapt_apply(ad, sums, tid, D);   // pmc_step_t.adapt_state: the proposal of the next step
    if (done_flag) {
        // the sums' host copy acknowledged before the completion word (agent-scope release: scaler_body.h says why that
        // is enough); the state updates are device memory, ordered for the next kernel by the kernel boundary
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
    }
    if (tid == 0) {
        *ticket = 0u;                     // ready for the next launch (the caller zeroes it once at allocation)
        if (done_flag) __hip_atomic_store(done_flag, done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ===========================================================================
// particle math: particles.py:215-231, tools.py:56-133, sampler.py:680-715
// ===========================================================================
__device__ __forceinline__ double np_logaddexp(double x, double y) {
    // numpy npy_logaddexp
    if (x == y) return x + 0.6931471805599453;
    const double tmp = x - y;
    if (tmp > 0) return x + log1p(exp(-tmp));
    else if (tmp <= 0) return y + log1p(exp(tmp));
    return tmp;   // NaN
}

__global__ __launch_bounds__(256) void logw_kernel(const double* __restrict__ logl, const double* __restrict__ beta,
                                                   const double* __restrict__ logz, double beta_final,
                                                   double* __restrict__ logw, int T, int64_t N) {
    const int64_t total = (int64_t)T * N;
    const double logT = log((double)T);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const double l = logl[e];
        double B;
        {
            B = l * beta[0] - logz[0];
            for (int i = 1; i < T; ++i) B = np_logaddexp(B, l * beta[i] - logz[i]);   // np.logaddexp.reduce(b, axis=0)
            logw[e] = l * beta_final - (B - logT);
        }
    }
}

__global__ __launch_bounds__(256) void max_partial_kernel(const double* __restrict__ v, int64_t P, double* __restrict__ part) {
    __shared__ double red[4];
    double m = -INFINITY;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < P; e += (int64_t)gridDim.x * 256) m = fmax(m, v[e]);
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

__global__ __launch_bounds__(256) void max_final_kernel(double* part, int nb) {
    if (threadIdx.x == 0) {
        double m = -INFINITY;
        for (int b = 0; b < nb; ++b) m = fmax(m, part[b]);
        part[nb] = m;
    }
}

__global__ __launch_bounds__(256) void wsum_partial_kernel(const double* __restrict__ logw, int64_t P, const double* __restrict__ mx,
                                                           const double* __restrict__ tot, int64_t kk, int mode,
                                                           double* __restrict__ part) {
    // mode 0: sum w, sum w^2 with w = exp(logw - max);  mode 1: sum 1-(1-w/tot)^k
    __shared__ double red[4];
    const double m = *mx;
    double s1 = 0.0, s2 = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < P; e += (int64_t)gridDim.x * 256) {
        const double w = exp(logw[e] - m);
        if (mode == 0) { s1 += w; s2 += w * w; }
        else { s1 += 1.0 - pow(1.0 - w / *tot, (double)kk); }
    }
    const double a = block_sum_256(s1, red, threadIdx.x);
    const double b = block_sum_256(s2, red, threadIdx.x);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = a; part[2 * blockIdx.x + 1] = b; }
}

__global__ void wsum_final_kernel(const double* __restrict__ part, int nb, double* __restrict__ out, int o1, int o2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < nb; ++i) { a += part[2 * i]; b += part[2 * i + 1]; }
        out[o1] = a;
        if (o2 >= 0) out[o2] = b;
    }
}

__global__ __launch_bounds__(256) void gather_kernel(const int64_t* __restrict__ idx, int64_t n_out, int D,
                                                     const double* __restrict__ u, const double* __restrict__ x,
                                                     const double* __restrict__ a, const double* __restrict__ b,
                                                     const double* __restrict__ c, double* __restrict__ uo,
                                                     double* __restrict__ xo, double* __restrict__ ao,
                                                     double* __restrict__ bo, double* __restrict__ co) {
    const int64_t total = n_out * D;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / D;
        const int j = (int)(e - r * D);
        const int64_t s = idx[r];
        if (u) uo[e] = u[s * D + j];
        if (x) xo[e] = x[s * D + j];
        if (j == 0) {
            if (a) ao[r] = a[s];
            if (b) bo[r] = b[s];
            if (c) co[r] = c[s];
        }
    }
}

// cumulative sum in the reference's order (np.cumsum / the running sum of
// tools.py:176-183 are sequential), so that resampled INDICES are bit-exact.
// np.cumsum adds left to right; a parallel scan would round differently, and the resampling indices are compared
// bit for bit with the reference's (searchsorted on this cdf).  So the additions stay one dependent chain -- but on
// LDS-resident chunks that the whole workgroup loads and stores coalesced (a lone thread walking global memory paid
// a memory round trip per element: 4 ms for a pool of 8e4, now 0.3 ms).
#define CUMSUM_CHUNK 4096
__global__ __launch_bounds__(256) void serial_cumsum_kernel(const double* __restrict__ w, int64_t P,
                                                            double* __restrict__ cdf, int normalise) {
    __shared__ double buf[CUMSUM_CHUNK];      // the chunk's weights
    __shared__ double acc[CUMSUM_CHUNK];      // its running sums (a second array: reads never wait for the writes)
    __shared__ double carry;
    if (threadIdx.x == 0) carry = 0.0;
    for (int64_t base = 0; base < P; base += CUMSUM_CHUNK) {
        const int m = (int)min((int64_t)CUMSUM_CHUNK, P - base);
        for (int i = threadIdx.x; i < m; i += 256) buf[i] = w[base + i];
        __syncthreads();
        if (threadIdx.x == 0) {
            double c = carry;
            int i = 0;
            for (; i + 16 <= m; i += 16) {
                double a[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) a[k] = buf[i + k];
#pragma unroll
                for (int k = 0; k < 16; ++k) { c += a[k]; a[k] = c; }
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[i + k] = a[k];
            }
            for (; i < m; ++i) { c += buf[i]; acc[i] = c; }
            carry = c;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += 256) cdf[base + i] = acc[i];
        __syncthreads();
    }
    (void)normalise;
}

__global__ __launch_bounds__(256) void scale_kernel(double* __restrict__ v, int64_t P) {
    const double last = v[P - 1];
    __syncthreads();
    // every block reads v[P-1] before anyone rewrites it: the last element is handled by
    // the thread that owns it, after its own read
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < P - 1; e += (int64_t)gridDim.x * 256) v[e] = v[e] / last;
}

__global__ void scale_last_kernel(double* __restrict__ v, int64_t P) {
    if (threadIdx.x == 0 && blockIdx.x == 0) v[P - 1] = v[P - 1] / v[P - 1];
}

__global__ __launch_bounds__(256) void searchsorted_kernel(const double* __restrict__ cdf, int64_t P,
                                                           const double* __restrict__ uniforms, double offset,
                                                           int64_t n_out, int systematic, int64_t* __restrict__ idx) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * 256) {
        double v;
        if (systematic) v = (offset + (double)i) / (double)n_out;          // tools.py:175
        else v = uniforms[i];
        int64_t lo = 0, hi = P;
        if (systematic) {
            // first j with cdf[j] >= v   ("while positions[i] > cumulative_sum: j += 1")
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (cdf[mid] < v) lo = mid + 1; else hi = mid; }
        } else {
            // searchsorted(side='right'): first j with cdf[j] > v
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (cdf[mid] <= v) lo = mid + 1; else hi = mid; }
        }
        idx[i] = lo < P ? lo : P - 1;
    }
}

// ===========================================================================
// host side
// ===========================================================================
static inline unsigned grid_for(int64_t n, int per_block, int cap = 2048) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

extern "C" int pmc_propose(int kind, const float* cur32, const double* cur64, const double* mu,
                           const double* inv_cov, const double* chol, double nu, double sigma, double cn_a,
                           const pmc_rng_t* rng, double* prop64, float* prop32, double* quad,
                           double* quad_prop, int64_t n, int32_t D, void* stream) {
    return pmc_propose_adapt(kind, cur32, cur64, mu, inv_cov, chol, nu, sigma, cn_a, rng, prop64, prop32, quad, quad_prop,
                             n, D, stream, nullptr);
}

int pmc_propose_adapt(int kind, const float* cur32, const double* cur64, const double* mu, const double* inv_cov,
                      const double* chol, double nu, double sigma, double cn_a, const pmc_rng_t* rng, double* prop64,
                      float* prop32, double* quad, double* quad_prop, int64_t n, int32_t D, void* stream,
                      const double* adapt) {
    if (n == 0) return 0;
    if ((!cur32) == (!cur64)) return pmc_fail("pmc_propose: exactly one of cur32 / cur64 must be given");
    if (!chol || !rng || n < 0 || D < 1) return pmc_fail("pmc_propose: bad argument");
    if (kind == PMC_KIND_TPCN && ((!mu && !adapt) || !inv_cov)) return pmc_fail("pmc_propose: tpCN needs mu and inv_cov");
    if (kind != PMC_KIND_TPCN && kind != PMC_KIND_RWM) return pmc_fail("pmc_propose: unknown kind");
    if (!prop64 && !prop32) return pmc_fail("pmc_propose: no output");
    {   // f64 matrix-core kernel (D <= 128); the LDS-staged VALU kernel below covers larger D
        const int rc = pmc_launch_propose_mfma(kind, cur32, cur64, mu, inv_cov, chol, nu, sigma, cn_a, rng, prop64,
                                               prop32, quad, quad_prop, n, D, (hipStream_t)stream, adapt);
        if (rc >= 0) return rc;
    }
    const size_t lds = (size_t)2 * D * (PROP_ROWS + 1) * sizeof(double);
    if (lds > 160 * 1024) return pmc_fail("pmc_propose: n_dim too large for the LDS-staged proposal");
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(propose_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pmc_fail_hip(e, "hipFuncSetAttribute(propose_kernel)");
    }
    hipLaunchKernelGGL(propose_kernel, dim3((unsigned)((n + PROP_ROWS - 1) / PROP_ROWS)), dim3(PROP_ROWS), lds,
                       (hipStream_t)stream, kind, cur32, cur64, mu, inv_cov, chol, nu, sigma, cn_a, *rng,
                       prop64, prop32, quad, quad_prop, n, (int)D, adapt);
    return pmc_check_launch("propose_kernel");
}

// ===========================================================================
// Philox variates of one step ahead of time (same values as the inline draws of propose_body.h / accept_kernel)
// ===========================================================================
__global__ __launch_bounds__(256) void rng_fill_kernel(pmc_rng_t rng, double gamma_shape, double* __restrict__ normal,
                                                       double* __restrict__ gamma, double* __restrict__ uniform,
                                                       int64_t n, int D) {
    const int pairs = (D + 1) >> 1;
    const int slots = pairs + 2;                         // per walker: D/2 normal pairs, the gamma, the uniform
    const int64_t total = n * slots;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t k = e / slots;
        const int sl = (int)(e - k * slots);
        const uint64_t gidx = rng.offset + (uint64_t)k;
        if (sl < pairs) {
            if (normal) {
                Philox ph(rng.seed, rng.step, gidx, 1);
                ph.ctr[0] = (uint32_t)sl;
                double a, b;
                ph.normal2(a, b);
                normal[k * D + 2 * sl] = a;
                if (2 * sl + 1 < D) normal[k * D + 2 * sl + 1] = b;
            }
        } else if (sl == pairs) {
            if (gamma && gamma_shape > 0.0) { Philox ph(rng.seed, rng.step, gidx, 0); gamma[k] = ph.std_gamma(gamma_shape); }
        } else if (uniform) {
            Philox ph(rng.seed, rng.step, gidx, 2);
            double ur, d;
            ph.uniform2(ur, d);
            uniform[k] = ur;
        }
    }
}

extern "C" int pmc_rng_fill(const pmc_rng_t* rng, double gamma_shape, double* normal, double* gamma, double* uniform,
                            int64_t n, int32_t D, void* stream) {
    if (!rng || n < 0 || D < 1) return pmc_fail("pmc_rng_fill: bad argument");
    if (n == 0) return 0;
    const int64_t total = n * (((D + 1) >> 1) + 2);
    hipLaunchKernelGGL(rng_fill_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, *rng, gamma_shape,
                       normal, gamma, uniform, n, (int)D);
    return pmc_check_launch("rng_fill_kernel");
}

static int check_scaler(const pmc_scaler_t* s) {
    if (!s || !s->low || !s->high || !s->kind || s->D < 1) return pmc_fail("pmc_scaler: bad descriptor");
    if (s->scale && (!s->mu || !s->sigma)) return pmc_fail("pmc_scaler: scale=1 needs mu and sigma");
    if (!s->log_width) return pmc_fail("pmc_scaler: log_width missing");
    return 0;
}

extern "C" int pmc_scaler_inverse(const pmc_scaler_t* s, const float* u_in, const double* u_in64, double* u_out,
                                  double* x, double* x_colmajor, double* logdetj, int32_t* finite, int64_t n,
                                  void* stream) {
    return pmc_scaler_inverse_prior(s, nullptr, u_in, u_in64, u_out, x, x_colmajor, logdetj, finite, nullptr, nullptr,
                                    nullptr, nullptr, n, stream);
}

extern "C" int pmc_scaler_inverse_prior(const pmc_scaler_t* s, const pmc_prior_t* prior, const float* u_in,
                                        const double* u_in64, double* u_out, double* x, double* x_colmajor,
                                        double* logdetj, int32_t* finite, double* logp, int32_t* finite_copy,
                                        double* logp_copy, const pmc_done_t* done, int64_t n, void* stream) {
    return pmc_scaler_inverse_prior_ex(s, prior, u_in, u_in64, u_out, x, x_colmajor, logdetj, finite, logp, finite_copy, logp_copy,
                                       done, n, stream, nullptr);
}

int pmc_scaler_inverse_prior_ex(const pmc_scaler_t* s, const pmc_prior_t* prior, const float* u_in,
                                const double* u_in64, double* u_out, double* x, double* x_colmajor,
                                double* logdetj, int32_t* finite, double* logp, int32_t* finite_copy,
                                double* logp_copy, const pmc_done_t* done, int64_t n, void* stream, const pmc_scaler_extra* extra) {
    if (int e = check_scaler(s)) return e;
    if (n == 0) return 0;
    if ((!u_in) == (!u_in64)) return pmc_fail("pmc_scaler_inverse: exactly one of u_in / u_in64 must be given");
    if (!u_out || !x || !logdetj || !finite || n < 0) return pmc_fail("pmc_scaler_inverse: bad argument");
    if ((prior != nullptr) != (logp != nullptr)) return pmc_fail("pmc_scaler_inverse_prior: prior and logp go together");
    if (logp_copy && !logp) return pmc_fail("pmc_scaler_inverse_prior: logp_copy without logp");
    if (prior && (!prior->family || !prior->loc || !prior->scale || prior->D != s->D))
        return pmc_fail("pmc_scaler_inverse_prior: bad prior descriptor");
    pmc_prior_t pr_val = {};
    if (prior) pr_val = *prior;
    pmc_scaler_extra ex{};
    if (extra && done && x_colmajor) ex = *extra;       // (the count travels with the completion word, the fill with the host copy)
    const size_t lds = (size_t)SCL_ROWS * s->D * sizeof(double) + SCL_ROWS * sizeof(int) +
                       ((x_colmajor || logp) ? (size_t)s->D * (SCL_ROWS + 1) * sizeof(double) : 0);
    if (lds > 160 * 1024 || s->D > 1024) return pmc_fail("pmc_scaler_inverse: n_dim too large");
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(scaler_inverse_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pmc_fail_hip(e, "hipFuncSetAttribute(scaler_inverse_kernel)");
    }
    hipLaunchKernelGGL(scaler_inverse_kernel, dim3((unsigned)((n + SCL_ROWS - 1) / SCL_ROWS)), dim3(SCL_THREADS), lds,
                       (hipStream_t)stream, *s, u_in, u_in64, u_out, x, x_colmajor, logdetj, finite, n, pr_val, logp, finite_copy,
                       logp_copy, done ? done->ticket : nullptr, done ? (long long*)done->flag : nullptr,
                       done ? (long long)done->value : 0LL, ex);
    return pmc_check_launch("scaler_inverse_kernel");
}

extern "C" int pmc_scaler_forward(const pmc_scaler_t* s, const double* x, double* u, int64_t n, void* stream) {
    if (int e = check_scaler(s)) return e;
    if (n == 0) return 0;
    if (!x || !u || n < 0) return pmc_fail("pmc_scaler_forward: bad argument");
    hipLaunchKernelGGL(scaler_forward_kernel, dim3(grid_for(n * s->D, 256)), dim3(256), 0, (hipStream_t)stream,
                       *s, x, u, n);
    return pmc_check_launch("scaler_forward_kernel");
}

extern "C" int64_t pmc_accept_workspace_bytes(int64_t n, int32_t D) {
    const int64_t nb = (n + ACC_ROWS - 1) / ACC_ROWS;
    return 64 + (nb > 0 ? nb : 1) * (int64_t)(D + 4) * (int64_t)sizeof(double);   // [ticket | partials]
}

static int accept_impl(int kind, int preconditioned, pmc_state_t* cur, const pmc_proposal_t* prop, double beta,
                       double nu, const pmc_rng_t* rng, double* alpha_out, int32_t* accept_out, double* sums,
                       double* sums_copy, bool armed, const pmc_done_t* done, void* workspace, int64_t n, int32_t D,
                       void* stream, const pmc_adapt_args* adapt = nullptr) {
    if (!cur || !prop || !rng || !sums || !workspace || n < 0 || D < 1) return pmc_fail("pmc_accept: bad argument");
    if (!cur->u || !cur->x || !cur->logdetj || !cur->logl || !cur->logp || !prop->u || !prop->x ||
        !prop->logdetj || !prop->logl || !prop->logp)
        return pmc_fail("pmc_accept: null state array");
    if (preconditioned && (!cur->theta32 || !cur->logdetj_flow || !prop->theta64 || !prop->logdetj_flow))
        return pmc_fail("pmc_accept: preconditioned kernels need theta and logdetj_flow");
    const int tpcn = (kind == PMC_KIND_TPCN);
    if (tpcn && (!prop->quad || !prop->quad_prop)) return pmc_fail("pmc_accept: tpCN needs the quadratic forms");
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)((n + ACC_ROWS - 1) / ACC_ROWS);
    unsigned* ticket = (unsigned*)workspace;
    double* partials = (double*)((char*)workspace + 64);
    if (nb == 0) {
        if (hipMemsetAsync(sums, 0, (size_t)(D + 4) * sizeof(double), st) != hipSuccess) return pmc_fail("pmc_accept: memset");
        return 0;
    }
    // the ticket word is re-armed by every call (a memset node ahead of the launch): no state between calls.
    // pmc_accept_armed skips it: the kernel leaves the word at zero itself.
    if (!armed && hipMemsetAsync(ticket, 0, sizeof(unsigned), st) != hipSuccess) return pmc_fail("pmc_accept: memset");
    hipLaunchKernelGGL(accept_kernel, dim3(nb), dim3(256), 0, st, preconditioned, tpcn, *cur, *prop, beta, nu, *rng,
                       alpha_out, accept_out, partials, ticket, sums, sums_copy, done ? (long long*)done->flag : nullptr,
                       done ? (long long)done->value : 0LL, n, (int)D, adapt ? *adapt : pmc_adapt_args{});
    return pmc_check_launch("accept_kernel");
}

extern "C" int pmc_accept(int kind, int preconditioned, pmc_state_t* cur, const pmc_proposal_t* prop, double beta,
                          double nu, const pmc_rng_t* rng, double* alpha_out, int32_t* accept_out, double* sums,
                          void* workspace, int64_t n, int32_t D, void* stream) {
    return accept_impl(kind, preconditioned, cur, prop, beta, nu, rng, alpha_out, accept_out, sums, nullptr, false,
                       nullptr, workspace, n, D, stream);
}

extern "C" int pmc_accept_armed(int kind, int preconditioned, pmc_state_t* cur, const pmc_proposal_t* prop, double beta,
                                double nu, const pmc_rng_t* rng, double* alpha_out, int32_t* accept_out, double* sums,
                                double* sums_copy, const pmc_done_t* done, void* workspace, int64_t n, int32_t D,
                                void* stream) {
    return accept_impl(kind, preconditioned, cur, prop, beta, nu, rng, alpha_out, accept_out, sums, sums_copy, true,
                       done, workspace, n, D, stream);
}

struct AdaptParts { const double* p[8]; };

__global__ __launch_bounds__(256) void adapt_update_kernel(AdaptParts parts, int n_parts, int D, double* total_out,
                                                           double* h_sums, pmc_adapt_args ad, long long* done_flag,
                                                           long long done_value) {
    __shared__ double tot[260];
    const int tid = threadIdx.x;
    for (int j = tid; j < D + 4; j += 256) {
        double t = parts.p[0][j];
        for (int k = 1; k < n_parts; ++k) t += parts.p[k][j];
        tot[j] = t;
        if (total_out) total_out[j] = t;
        if (h_sums) h_sums[j] = t;
    }
    __syncthreads();
    if (ad.state && ad.mode) adapt_apply(ad, tot, tid, D);
    if (done_flag) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done_flag, done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

extern "C" int pmc_adapt_update(const double* const* parts, int32_t n_parts, int32_t D, double* total_out, double* h_sums,
                                double* adapt_state, int32_t adapt_mode, double c_sigma, double c_mu, double cap,
                                double n_total, const pmc_done_t* done, void* stream) {
    if (!parts || n_parts < 1 || n_parts > 8 || D < 1 || D > 256) return pmc_fail("pmc_adapt_update: bad argument");
    AdaptParts ap{};
    for (int k = 0; k < n_parts; ++k) {
        if (!parts[k]) return pmc_fail("pmc_adapt_update: null part");
        ap.p[k] = parts[k];
    }
    pmc_adapt_args ad{adapt_state, adapt_state ? adapt_mode : 0, c_sigma, c_mu, cap, n_total, {}, 0};
    hipLaunchKernelGGL(adapt_update_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ap, (int)n_parts, (int)D, total_out,
                       h_sums, ad, done ? (long long*)done->flag : nullptr, done ? (long long)done->value : 0LL);
    return pmc_check_launch("adapt_update_kernel");
}

// ---------------------------------------------------------------------------------------------------------------------
// The one exchange between the ranks of a sharded step (mcmc.py:152-156 need the GLOBAL sums: SURVEY.md 8(e)): an
// all-reduce of D + 4 doubles, as a launch of this library on the step's own stream -- so that the sharded step runs
// behind pmc_pipeline_* like the single-rank one, without a host round trip or a torch.distributed call between the last
// accept and the adaptation.  One GPU per rank, all on one node: every rank owns a MAILBOX in its HBM (uncached
// allocation), shared with the others through hipIpcGetMemHandle / hipIpcOpenMemHandle (xGMI peer access), laid out
//     [2 parities][world][width doubles | sequence word].
// Call number `seq` (all ranks call in the same order): a rank adds its own parts, writes the total into slot
// [seq & 1][rank] of EVERY mailbox (its own too) with system-scope stores, then the sequence word of that slot with a
// system-scope release; it then waits until the `world` words of its own mailbox show `seq` and adds the slots IN RANK ORDER
// -- every rank forms the same sum bit for bit, whatever the arrival order.  A slot is rewritten two calls later, by when
// every reader of the old value has published its next call (which it does only after reading).  The adaptation and the
// completion word follow in the same kernel (adapt_update_kernel's tail).  The wait is bounded: on a timeout the sums'
// first entry is NaN and the completion word is written all the same -- the host raises.
// ---------------------------------------------------------------------------------------------------------------------
struct pmc_comm {
    int rank, world, width, stride;        // stride: doubles per slot (width + 1, rounded to 8)
    double* own;                           // [2][world][stride]: uncached device memory (kind 0) or the device address of host_map[rank] (kind 1)
    double* peer[8];                       // the ranks' mailboxes as mapped into this process (peer[rank] == own)
    long long seq;
    hipIpcMemHandle_t handle;              // kind 0: the IPC handle; kind 1: the name of the POSIX shared-memory object (a C string)
    bool connected;
    int kind;                              // 0: device memory shared through hipIpc (xGMI peer stores); 1: pinned host memory (PCIe)
    size_t bytes;
    void* host_map[8];                     // kind 1: the ranks' segments as mmap()ed here (registered with the HIP runtime)
};

struct CommPeers { double* p[8]; };

__global__ __launch_bounds__(256) void comm_adapt_kernel(AdaptParts parts, int n_parts, int D, CommPeers peers, double* own, int rank,
                                                         int world, int stride, long long seq, long long timeout_ticks,
                                                         double* total_out, double* h_sums, pmc_adapt_args ad, long long* done_flag,
                                                         long long done_value) {
    __shared__ double tot[260];
    __shared__ int late;
    const int tid = threadIdx.x, W = D + 4;
    const int par = (int)(seq & 1);
    if (tid == 0) late = 0;
    for (int j = tid; j < W; j += 256) {
        double t = parts.p[0][j];
        for (int k = 1; k < n_parts; ++k) t += parts.p[k][j];
        for (int r = 0; r < world; ++r)
            __hip_atomic_store(peers.p[r] + ((size_t)par * world + rank) * stride + j, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if (tid < world)
        __hip_atomic_store(reinterpret_cast<long long*>(peers.p[tid] + ((size_t)par * world + rank) * stride + (stride - 1)), seq,
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (tid < world) {
        const long long* word = reinterpret_cast<const long long*>(own + ((size_t)par * world + tid) * stride + (stride - 1));
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            __builtin_amdgcn_s_sleep(8);
            if (timeout_ticks > 0 && wall_clock64() - t0 > timeout_ticks) { late = 1; break; }
        }
    }
    __syncthreads();
    for (int j = tid; j < W; j += 256) {
        double t = 0.0;
        for (int r = 0; r < world; ++r)
            t += __hip_atomic_load(own + ((size_t)par * world + r) * stride + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (late && j == 0) t = __builtin_nan("");
        tot[j] = t;
        if (total_out) total_out[j] = t;
        if (h_sums) h_sums[j] = t;
    }
    __syncthreads();
    if (ad.state && ad.mode && !late) adapt_apply(ad, tot, tid, D);
    if (done_flag) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done_flag, done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static pmc_comm* comm_new(int32_t rank, int32_t world, int32_t width, int kind, const char* who) {
    if (world < 1 || world > 8 || rank < 0 || rank >= world || width < 1 || width > 260) { pmc_fail(who); return nullptr; }
    pmc_comm* c = new pmc_comm();
    c->rank = rank; c->world = world; c->width = width; c->stride = (width + 1 + 7) & ~7;
    c->seq = 0; c->connected = false; c->kind = kind; c->own = nullptr;
    c->bytes = (size_t)2 * world * c->stride * sizeof(double);
    for (int r = 0; r < 8; ++r) { c->peer[r] = nullptr; c->host_map[r] = nullptr; }
    memset(&c->handle, 0, sizeof(c->handle));
    return c;
}

extern "C" void* pmc_comm_create(int32_t rank, int32_t world, int32_t width) {
    pmc_comm* c = comm_new(rank, world, width, 0, "pmc_comm_create: 1..8 ranks, width <= 260");
    if (!c) return nullptr;
    void* ptr = nullptr;
    // uncached: a peer's stores over xGMI must be what this device's loads see, without a cache line of its own in between.
    // No cached fallback: the kernel's protocol is only validated for memory the device does not cache -- a caller that
    // cannot have it takes the host mailboxes (pmc_comm_create_host) or its process group's all-reduce instead.
    hipError_t e = hipExtMallocWithFlags(&ptr, c->bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) { (void)hipGetLastError(); pmc_fail_hip(e, "pmc_comm_create: uncached allocation of the mailbox"); delete c; return nullptr; }
    if (hipMemset(ptr, 0, c->bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { pmc_fail("pmc_comm_create: hipMemset"); (void)hipFree(ptr); delete c; return nullptr; }
    c->own = (double*)ptr;
    c->peer[rank] = c->own;
    if (world > 1) {
        e = hipIpcGetMemHandle(&c->handle, ptr);
        if (e != hipSuccess) { pmc_fail_hip(e, "pmc_comm_create: hipIpcGetMemHandle"); (void)hipFree(ptr); delete c; return nullptr; }
    } else c->connected = true;
    return c;
}

// kind 1: the mailbox in pinned, coherent HOST memory -- a POSIX shared-memory object every rank maps and registers with
// its HIP runtime (hipHostRegister: fine-grained, never cached by a device).  The same kernel and protocol; every store and
// every poll crosses PCIe instead of xGMI.  The tier for nodes where hipIpc peer mappings are not to be had, and the way
// the protocol is exercised over a non-local, uncached path on a single GPU (tests/test_gpu_sharded_sampler.py).
static void* comm_map_shm(const char* name, size_t bytes, bool create, void** dev_out, const char* who) {
    const int fd = shm_open(name, create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) { pmc_fail(who); return nullptr; }
    if (create && ftruncate(fd, (off_t)bytes) != 0) { close(fd); shm_unlink(name); pmc_fail(who); return nullptr; }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { if (create) shm_unlink(name); pmc_fail(who); return nullptr; }
    if (create) memset(p, 0, bytes);
    hipError_t e = hipHostRegister(p, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    if (e == hipSuccess) e = hipHostGetDevicePointer(dev_out, p, 0);
    if (e != hipSuccess) { (void)hipGetLastError(); munmap(p, bytes); if (create) shm_unlink(name); pmc_fail_hip(e, who); return nullptr; }
    return p;
}

extern "C" void* pmc_comm_create_host(int32_t rank, int32_t world, int32_t width) {
    pmc_comm* c = comm_new(rank, world, width, 1, "pmc_comm_create_host: 1..8 ranks, width <= 260");
    if (!c) return nullptr;
    static int counter = 0;
    char* name = reinterpret_cast<char*>(&c->handle);
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    snprintf(name, 64, "/pmc_comm_%d_%d_%lx", (int)getpid(), counter++, (unsigned long)ts.tv_nsec);
    void* dev = nullptr;
    void* p = comm_map_shm(name, c->bytes, true, &dev, "pmc_comm_create_host: shared-memory mailbox");
    if (!p) { delete c; return nullptr; }
    c->host_map[rank] = p;
    c->own = (double*)dev;
    c->peer[rank] = c->own;
    if (world == 1) c->connected = true;
    return c;
}

extern "C" int pmc_comm_handle(void* cc, void* out64) {
    pmc_comm* c = (pmc_comm*)cc;
    if (!c || !out64) return pmc_fail("pmc_comm_handle: null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    memcpy(out64, &c->handle, 64);
    return 0;
}

// handles: world x 64 bytes, rank r's at 64 r (as pmc_comm_handle returned them in the ranks' processes)
extern "C" int pmc_comm_connect(void* cc, const void* handles) {
    pmc_comm* c = (pmc_comm*)cc;
    if (!c || !handles) return pmc_fail("pmc_comm_connect: null argument");
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        if (c->kind == 1) {
            char name[64];
            memcpy(name, (const char*)handles + 64 * r, 64);
            name[63] = 0;
            void* dev = nullptr;
            void* p = comm_map_shm(name, c->bytes, false, &dev, "pmc_comm_connect: a peer's shared-memory mailbox");
            if (!p) return 1;
            c->host_map[r] = p;
            c->peer[r] = (double*)dev;
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)handles + 64 * r, 64);
        void* ptr = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return pmc_fail_hip(e, "pmc_comm_connect: hipIpcOpenMemHandle");
        c->peer[r] = (double*)ptr;
    }
    c->connected = true;
    return 0;
}

extern "C" void pmc_comm_destroy(void* cc) {
    pmc_comm* c = (pmc_comm*)cc;
    if (!c) return;
    if (c->kind == 1) {
        for (int r = 0; r < c->world; ++r)
            if (c->host_map[r]) { (void)hipHostUnregister(c->host_map[r]); munmap(c->host_map[r], c->bytes); }
        if (c->host_map[c->rank] && reinterpret_cast<const char*>(&c->handle)[0]) shm_unlink(reinterpret_cast<const char*>(&c->handle));
        delete c;
        return;
    }
    for (int r = 0; r < c->world; ++r)
        if (r != c->rank && c->peer[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
    if (c->own) (void)hipFree(c->own);
    delete c;
}

// 0: device mailboxes behind hipIpc handles, 1: host mailboxes in POSIX shared memory
extern "C" int pmc_comm_kind(void* cc) { return cc ? ((pmc_comm*)cc)->kind : -1; }

// host mailboxes: once EVERY rank has connected (the caller knows: it gathered the ranks' pmc_comm_connect results) the name
// can go -- the mappings stay valid, and a process that dies later leaves nothing behind in /dev/shm.  No-op for kind 0.
extern "C" int pmc_comm_unlink(void* cc) {
    pmc_comm* c = (pmc_comm*)cc;
    if (!c) return pmc_fail("pmc_comm_unlink: null communicator");
    if (c->kind == 1 && c->host_map[c->rank]) {
        char* name = reinterpret_cast<char*>(&c->handle);
        if (name[0]) { shm_unlink(name); name[0] = 0; }
    }
    return 0;
}

// total over the parts of this rank AND over the ranks (rank order), then as pmc_adapt_update: total_out / h_sums / the
// adaptation / done.  Every rank must make the same sequence of calls.  timeout_s <= 0: wait for ever.
extern "C" int pmc_comm_adapt_update(void* cc, const double* const* parts, int32_t n_parts, int32_t D, double* total_out,
                                     double* h_sums, double* adapt_state, int32_t adapt_mode, double c_sigma, double c_mu, double cap,
                                     double n_total, const pmc_done_t* done, double timeout_s, void* stream) {
    pmc_comm* c = (pmc_comm*)cc;
    if (!c || !c->connected) return pmc_fail("pmc_comm_adapt_update: the communicator is not connected");
    if (!parts || n_parts < 1 || n_parts > 8 || D < 1 || D + 4 > c->width) return pmc_fail("pmc_comm_adapt_update: bad argument");
    AdaptParts ap{};
    for (int k = 0; k < n_parts; ++k) {
        if (!parts[k]) return pmc_fail("pmc_comm_adapt_update: null part");
        ap.p[k] = parts[k];
    }
    CommPeers pe{};
    for (int r = 0; r < c->world; ++r) pe.p[r] = c->peer[r];
    pmc_adapt_args ad{adapt_state, adapt_state ? adapt_mode : 0, c_sigma, c_mu, cap, n_total, {}, 0};
    c->seq += 1;
    const long long ticks = timeout_s > 0.0 ? (long long)(timeout_s * 1.0e8) : 0LL;        // wall_clock64: 100 MHz
    hipLaunchKernelGGL(comm_adapt_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ap, (int)n_parts, (int)D, pe, c->own, c->rank,
                       c->world, c->stride, c->seq, ticks, total_out, h_sums, ad, done ? (long long*)done->flag : nullptr,
                       done ? (long long)done->value : 0LL);
    return pmc_check_launch("comm_adapt_kernel");
}

int pmc_accept_adapt(int kind, int preconditioned, pmc_state_t* cur, const pmc_proposal_t* prop, double beta, double nu,
                     const pmc_rng_t* rng, double* alpha_out, int32_t* accept_out, double* sums, double* sums_copy,
                     const pmc_done_t* done, void* workspace, int64_t n, int32_t D, void* stream,
                     const pmc_adapt_args* adapt) {
    if (adapt && ((adapt->state && adapt->mode) || adapt->n_other) && D > 256)
        return pmc_fail("pmc_accept: device adaptation needs D <= 256");
    return accept_impl(kind, preconditioned, cur, prop, beta, nu, rng, alpha_out, accept_out, sums, sums_copy, true,
                       done, workspace, n, D, stream, adapt);
}

extern "C" int pmc_logw(const double* logl, const double* beta, const double* logz, double beta_final,
                        double* logw, int32_t T, int64_t N, void* stream) {
    if (!logl || !beta || !logz || !logw || T < 1 || N < 0) return pmc_fail("pmc_logw: bad argument");
    if (N == 0) return 0;
    hipLaunchKernelGGL(logw_kernel, dim3(grid_for((int64_t)T * N, 256)), dim3(256), 0, (hipStream_t)stream,
                       logl, beta, logz, beta_final, logw, (int)T, N);
    return pmc_check_launch("logw_kernel");
}

#define RED_BLOCKS 256
extern "C" int64_t pmc_reduce_workspace_bytes(int64_t P) {
    (void)P;
    return (int64_t)(2 * RED_BLOCKS + 8) * (int64_t)sizeof(double);
}

extern "C" int pmc_logw_stats(const double* logw, int64_t P, int64_t k, double* stats, void* workspace,
                              void* stream) {
    if (!logw || !stats || !workspace || P < 1) return pmc_fail("pmc_logw_stats: bad argument");
    hipStream_t st = (hipStream_t)stream;
    double* ws = (double*)workspace;
    const int nb = (int)grid_for(P, 256, RED_BLOCKS);
    hipLaunchKernelGGL(max_partial_kernel, dim3(nb), dim3(256), 0, st, logw, P, ws);
    hipLaunchKernelGGL(max_final_kernel, dim3(1), dim3(256), 0, st, ws, nb);
    (void)hipMemcpyAsync(stats, ws + nb, sizeof(double), hipMemcpyDeviceToDevice, st);
    hipLaunchKernelGGL(wsum_partial_kernel, dim3(nb), dim3(256), 0, st, logw, P, (const double*)stats,
                       (const double*)nullptr, (int64_t)0, 0, ws);
    hipLaunchKernelGGL(wsum_final_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, nb, stats, 1, 2);
    if (k > 0) {
        hipLaunchKernelGGL(wsum_partial_kernel, dim3(nb), dim3(256), 0, st, logw, P, (const double*)stats,
                           (const double*)(stats + 1), k, 1, ws);
        hipLaunchKernelGGL(wsum_final_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, nb, stats, 3, -1);
    }
    return pmc_check_launch("pmc_logw_stats");
}

extern "C" int pmc_gather(const int64_t* idx, int64_t n_out, int32_t D, const double* u, const double* x,
                          const double* logdetj, const double* logl, const double* logp, double* u_out,
                          double* x_out, double* logdetj_out, double* logl_out, double* logp_out, void* stream) {
    if (!idx || n_out < 0 || D < 1) return pmc_fail("pmc_gather: bad argument");
    if (n_out == 0) return 0;
    hipLaunchKernelGGL(gather_kernel, dim3(grid_for(n_out * D, 256)), dim3(256), 0, (hipStream_t)stream, idx,
                       n_out, (int)D, u, x, logdetj, logl, logp, u_out, x_out, logdetj_out, logl_out, logp_out);
    return pmc_check_launch("gather_kernel");
}

extern "C" int pmc_resample_multinomial(const double* w, int64_t P, const double* uniforms, int64_t n_out,
                                        double* cdf, int64_t* idx, void* stream) {
    if (!w || !uniforms || !cdf || !idx || P < 1 || n_out < 0) return pmc_fail("pmc_resample_multinomial: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(serial_cumsum_kernel, dim3(1), dim3(256), 0, st, w, P, cdf, 1);
    // cdf /= cdf[-1]  (numpy's legacy choice): all but the last element first, then the last
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(P, 256, 256)), dim3(256), 0, st, cdf, P);
    hipLaunchKernelGGL(scale_last_kernel, dim3(1), dim3(64), 0, st, cdf, P);
    if (n_out > 0)
        hipLaunchKernelGGL(searchsorted_kernel, dim3(grid_for(n_out, 256)), dim3(256), 0, st, (const double*)cdf, P,
                           uniforms, 0.0, n_out, 0, idx);
    return pmc_check_launch("pmc_resample_multinomial");
}

extern "C" int pmc_resample_systematic(const double* w, int64_t P, double offset, int64_t n_out, double* cdf,
                                       int64_t* idx, void* stream) {
    if (!w || !cdf || !idx || P < 1 || n_out < 0) return pmc_fail("pmc_resample_systematic: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(serial_cumsum_kernel, dim3(1), dim3(256), 0, st, w, P, cdf, 0);
    if (n_out > 0)
        hipLaunchKernelGGL(searchsorted_kernel, dim3(grid_for(n_out, 256)), dim3(256), 0, st, (const double*)cdf, P,
                           (const double*)nullptr, offset, n_out, 1, idx);
    return pmc_check_launch("pmc_resample_systematic");
}

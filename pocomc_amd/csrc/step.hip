// Composite entry points of one MCMC step (pocomc/mcmc.py:74-149): everything the device does
// before and after the host's prior / likelihood call, enqueued by ONE C call each, so that the
// host-side driver spends its time in the user's black boxes instead of in dispatch overhead.
//
//   pmc_step_pre :  [H2D mu] -> propose -> flow inverse (one launch for the affine flows) -> scaler inverse [+ prior] -> D2H x', finite
//   pmc_step_post:  H2D logl', logp' -> accept + reductions -> D2H sums
//
// Pure sequencing of the single-purpose entry points (same kernels, same stream order); host
// buffers must be pinned for the copies to be asynchronous.

#include <hip/hip_runtime.h>
#include <time.h>
#include "pmc_internal.h"
#include "scaler_body.h"

#ifdef PMC_DEBUG_HOOKS                               // measurement builds only (make DEBUG_HOOKS=1): not in the product library
static long long* g_epilogue_stamps = nullptr;       // pmc_debug_set_epilogue_stamps
static int64_t g_epilogue_stamps_n = 0;
#endif

extern "C" int pmc_step_pre(const pmc_step_t* s, const pmc_rng_t* rng, double nu, double sigma, double cn_a,
                            void* stream) {
    if (!s || !rng) return pmc_fail("pmc_step_pre: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = s->n;
    const int32_t D = s->D;
    const bool tpcn = (s->kind == PMC_KIND_TPCN);
    // throughput mode: this step's Philox variates were drawn ahead of time (see pmc_step_t.rng_normal)
    pmc_rng_t rr = *rng;
    const bool prefill = s->rng_ready && s->rng_normal[0] && s->rng_normal[1] && s->rng_uniform[0] &&
                         s->rng_uniform[1] && (!tpcn || (s->rng_gamma[0] && s->rng_gamma[1])) && !rng->normal &&
                         !rng->gamma && !rng->uniform;
    const int rb = (int)(rng->step & 1);
    const double gshape = tpcn ? 0.5 * ((double)D + nu) : 0.0;             // mcmc.py:80
    if (prefill) {
        if (*s->rng_ready != (int64_t)rng->step) {                       // first step of a run
            int rcf = pmc_rng_fill(rng, gshape, s->rng_normal[rb], s->rng_gamma[rb], s->rng_uniform[rb], n, D, stream);
            if (rcf) return rcf;
        }
        rr.normal = s->rng_normal[rb];
        rr.gamma = tpcn ? s->rng_gamma[rb] : nullptr;
        rng = &rr;
    }
    // what follows the last kernel whose results the host waits for
    auto finish = [&]() -> int {
        if (s->ev_pre_done) (void)hipEventRecord((hipEvent_t)s->ev_pre_done, st);
        if (prefill) {
            pmc_rng_t nx = *rng;
            nx.normal = nullptr; nx.gamma = nullptr; nx.uniform = nullptr;
            nx.step = rng->step + 1;
            int rcf = pmc_rng_fill(&nx, gshape, s->rng_normal[rb ^ 1], s->rng_gamma[rb ^ 1], s->rng_uniform[rb ^ 1], n, D,
                                   stream);
            if (rcf) return rcf;
            *s->rng_ready = (int64_t)nx.step;
        }
        return 0;
    };
    // host_direct: the kernels read mu from / write x', finite, logp' to pinned host memory themselves
    const bool direct = s->host_direct && !s->p_xT;
    const double* mu = s->mu;
    // adaptation on the device: sigma, cn_a and mu come from adapt_state (pmc_step_post keeps it up to date)
    const double* adapt = (s->adapt_state && s->adapt_mode) ? s->adapt_state : nullptr;
    if (!adapt && tpcn && s->h_mu) {
        if (direct) mu = s->h_mu;
        else if (hipMemcpyAsync((void*)s->mu, s->h_mu, (size_t)D * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess)
            return pmc_fail("pmc_step_pre: H2D mu");
    }
    int rc;
    bool fused = false;
    int scaled = 0;                                  // the sweep's epilogue applied the scaler (+ prior) as well
    // scaler inverse and (when it runs on the device) Prior.logpdf: one launch, or the epilogue of the fused sweep
    const pmc_prior_t* pr = s->prior;
    double* lp = pr ? s->p_logp : nullptr;
    double* xT = direct ? s->h_x : s->p_xT;
    int32_t* fin2 = direct ? s->h_fin : nullptr;
    double* lp2 = (direct && pr) ? s->h_logp_out : nullptr;
    pmc_done_t dn{s->h_done, (int64_t)rng->step + 1, s->done_ticket};
    const pmc_done_t* done = (direct && s->h_done && s->done_ticket) ? &dn : nullptr;
    if (s->preconditioned && !(s->no_fuse & 1) && !pmc_tri6_preferred(s->maf) &&
        (s->inverse_algo == PMC_INVERSE_AUTO || s->inverse_algo == PMC_INVERSE_TRIANGULAR)) {
        // proposal + flow inverse (+ scaler) in one launch (affine flows, D <= 64, fewer than 16 hidden tiles).  The wider
        // flows take the lane-per-walker sweep with the proposal and the scaler as launches of their own: round 5 built the
        // fused instances of that sweep (proposal prologue up to D = 128, scaler / prior / x' epilogue, float32 and 16-bit
        // helpers) and measured them -- the proposal (~90 us per wavefront at D = 128) and the scaler (~40 us) are latency
        // chains that cost the same inside the sweep's launch as in their own, and the float32 five-wavefront instance has
        // no registers for them: config 5 255 (fused) against 300 steps/s, 16-bit helpers 420 against 415 (DESIGN.md
        // appendix A); no_fuse & 2 keeps the scaler apart
        ScalerEpi epi{};
        const bool want_epi = !(s->no_fuse & 2) && s->scaler && s->scaler->low && s->scaler->high && s->scaler->kind &&
                              s->scaler->log_width && (!s->scaler->scale || (s->scaler->mu && s->scaler->sigma)) &&
                              (!pr || (pr->family && pr->loc && pr->scale && pr->D == D));
        if (want_epi) {
            epi.s = *s->scaler;
            epi.have_prior = pr ? 1 : 0;
            if (pr) epi.pr = *pr;
            epi.u_out = s->p_u; epi.x_out = s->p_x; epi.x_colmajor = xT; epi.ldj_out = s->p_logdetj;
            epi.finite_out = s->p_fin; epi.logp_out = lp; epi.finite_copy = fin2; epi.logp_copy = lp2;
            epi.done_ticket = done ? done->ticket : nullptr;
            epi.done_flag = done ? (long long*)done->flag : nullptr;
            epi.done_value = done ? (long long)done->value : 0LL;
            epi.bad_count = (done && s->h_clean) ? s->clean_count : nullptr;
            epi.bad_flag = (done && s->h_clean && s->clean_count) ? (long long*)s->h_clean : nullptr;
#ifdef PMC_DEBUG_HOOKS
            epi.stamps = (n == g_epilogue_stamps_n) ? g_epilogue_stamps : nullptr;
#endif
            epi.fill_x = (s->fill_rejected && epi.bad_flag && xT) ? s->cur.x : nullptr;
        }
        if (s->ev_inv0) (void)hipEventRecord((hipEvent_t)s->ev_inv0, st);
        rc = pmc_launch_propose_inverse_tri4(s->kind, s->cur.theta32, mu, s->inv_cov, s->chol, nu, sigma, cn_a, rng,
                                             s->p_theta64, tpcn ? s->quad : nullptr, tpcn ? s->p_quad : nullptr, s->maf,
                                             s->p_u32, s->p_ldjf, n, st, adapt, want_epi ? &epi : nullptr, &scaled);
        if (rc > 0) return rc;
        fused = (rc == 0);
        if (!fused) scaled = 0;
        if (fused && s->ev_inv1) (void)hipEventRecord((hipEvent_t)s->ev_inv1, st);
    }
    if (!fused) {
        rc = pmc_propose_adapt(s->kind, s->preconditioned ? s->cur.theta32 : nullptr,
                               s->preconditioned ? nullptr : s->cur.u, mu, s->inv_cov, s->chol, nu, sigma, cn_a, rng,
                               s->p_theta64, s->preconditioned ? s->p_theta32 : nullptr, tpcn ? s->quad : nullptr,
                               tpcn ? s->p_quad : nullptr, n, D, stream, adapt);
        if (rc) return rc;
    }
    // the scaler launch of its own counts the rows that do not reach the likelihood like the fused epilogue does (and fills
    // their host rows when asked) whenever it hands x' to the host itself and evaluates the prior
    pmc_scaler_extra sx{};
    const bool counted = done && s->h_clean && s->clean_count && xT && (scaled || pr);
    if (!scaled && counted) { sx.bad_count = s->clean_count; sx.bad_flag = (long long*)s->h_clean; sx.fill_x = s->fill_rejected ? s->cur.x : nullptr; }
    if (s->h_clean && !counted) *s->h_clean = -1;     // (this launch sequence does not count)
    if (scaled) {
        rc = 0;
    } else if (s->preconditioned) {
        if (!fused) {
            if (s->ev_inv0) (void)hipEventRecord((hipEvent_t)s->ev_inv0, st);
            rc = pmc_maf_inverse(s->maf, s->p_theta32, s->p_u32, s->p_ldjf, n, s->inverse_algo, stream);
            if (s->ev_inv1) (void)hipEventRecord((hipEvent_t)s->ev_inv1, st);
            if (rc) return rc;
        }
        rc = pmc_scaler_inverse_prior_ex(s->scaler, pr, s->p_u32, nullptr, s->p_u, s->p_x, xT, s->p_logdetj, s->p_fin, lp,
                                         fin2, lp2, done, n, stream, &sx);
    } else {
        rc = pmc_scaler_inverse_prior_ex(s->scaler, pr, nullptr, s->p_theta64, s->p_u, s->p_x, xT, s->p_logdetj, s->p_fin,
                                         lp, fin2, lp2, done, n, stream, &sx);
    }
    if (rc) return rc;
    if (direct) return finish();
    if (pr) {
        if (hipMemcpyAsync(s->h_logp_out, s->p_logp, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess)
            return pmc_fail("pmc_step_pre: D2H logp");
    }
    const double* xsrc = s->p_xT ? s->p_xT : s->p_x;
    if (hipMemcpyAsync(s->h_x, xsrc, (size_t)n * D * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(s->h_fin, s->p_fin, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess)
        return pmc_fail("pmc_step_pre: D2H");
    return finish();
}

extern "C" int pmc_propose_inverse(int kind, const float* cur32, const double* mu, const double* inv_cov,
                                   const double* chol, double nu, double sigma, double cn_a, const pmc_rng_t* rng,
                                   double* prop64, double* quad, double* quad_prop, const pmc_maf_t* maf, float* u_out,
                                   float* ladj, int64_t n, void* stream) {
    if (!cur32 || !chol || !rng || !prop64 || !maf || !u_out || n < 0) return pmc_fail("pmc_propose_inverse: bad argument");
    if (kind == PMC_KIND_TPCN && (!mu || !inv_cov || !quad || !quad_prop))
        return pmc_fail("pmc_propose_inverse: tpCN needs mu, inv_cov and the quadratic-form outputs");
    if (n == 0) return 0;
    const int rc = pmc_launch_propose_inverse_tri4(kind, cur32, mu, inv_cov, chol, nu, sigma, cn_a, rng, prop64, quad,
                                                   quad_prop, maf, u_out, ladj, n, (hipStream_t)stream);
    if (rc < 0) return pmc_fail("pmc_propose_inverse: only the affine flows with D <= 64 have a fused instance");
    return rc;
}

extern "C" int pmc_step_post(const pmc_step_t* s, const pmc_rng_t* rng, double beta, double nu, int want_mask,
                             int copy_sums, void* stream) {
    if (!s || !rng) return pmc_fail("pmc_step_post: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = s->n;
    pmc_rng_t rr = *rng;
    if (s->rng_ready && s->rng_uniform[0] && s->rng_uniform[1] && !rng->uniform && !rng->normal && !rng->gamma &&
        *s->rng_ready == (int64_t)rng->step + 1) {
        rr.uniform = s->rng_uniform[rng->step & 1];                      // drawn ahead of time by pmc_step_pre
        rng = &rr;
    }
    const bool direct = s->host_direct && !s->p_xT;
    if (!direct) {
        if (hipMemcpyAsync(s->p_logl, s->h_logl, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess)
            return pmc_fail("pmc_step_post: H2D");
        if (!s->prior &&     // with a device prior logp' never left the device
            hipMemcpyAsync(s->p_logp, s->h_logp, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess)
            return pmc_fail("pmc_step_post: H2D");
    }
    pmc_state_t cur = s->cur;
    pmc_proposal_t prop;
    prop.theta64 = s->preconditioned ? s->p_theta64 : nullptr;
    prop.u = s->p_u; prop.x = s->p_x; prop.logdetj = s->p_logdetj;
    // host_direct: the accept kernel reads logl' (and a host-evaluated logp') from the pinned host buffers
    prop.logl = direct ? s->h_logl : s->p_logl;
    prop.logp = (direct && !s->prior) ? s->h_logp : s->p_logp;
    prop.logdetj_flow = s->preconditioned ? s->p_ldjf : nullptr;
    prop.quad = s->quad; prop.quad_prop = s->p_quad;
    int rc;
    if (direct) {
        pmc_done_t dn{s->h_done ? s->h_done + 1 : nullptr, (int64_t)rng->step + 1, nullptr};
        pmc_adapt_args ad{s->adapt_state, s->adapt_state ? s->adapt_mode : 0, s->adapt_c_sigma, s->adapt_c_mu,
                          s->adapt_cap, s->adapt_n_total, {}, 0};
        if (s->adapt_n_other < 0 || s->adapt_n_other > 7) return pmc_fail("pmc_step_post: adapt_n_other out of range");
        for (int k = 0; k < s->adapt_n_other; ++k) {
            if (!s->adapt_other[k]) return pmc_fail("pmc_step_post: null adapt_other");
            ad.other[k] = s->adapt_other[k];
        }
        ad.n_other = s->adapt_n_other;
        rc = pmc_accept_adapt(s->kind, s->preconditioned, &cur, &prop, beta, nu, rng, s->alpha, s->accept, s->sums,
                              copy_sums ? s->h_sums : nullptr, (copy_sums && s->h_done) ? &dn : nullptr, s->ws, n, s->D,
                              stream, &ad);
    }
    else if (s->adapt_state && s->adapt_mode)
        return pmc_fail("pmc_step_post: adaptation on the device needs host_direct");
    else
        rc = pmc_accept(s->kind, s->preconditioned, &cur, &prop, beta, nu, rng, s->alpha, s->accept, s->sums, s->ws, n,
                        s->D, stream);
    if (rc) return rc;
    if (copy_sums && !direct &&
        hipMemcpyAsync(s->h_sums, s->sums, (size_t)(s->D + 4) * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess)
        return pmc_fail("pmc_step_post: D2H sums");
    if (want_mask && s->h_accept &&
        hipMemcpyAsync(s->h_accept, s->accept, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess)
        return pmc_fail("pmc_step_post: D2H accept");
    return 0;
}

#ifdef PMC_DEBUG_HOOKS
// measurement only (bench.py, PMC_BENCH_EPI_STAMPS, a library built with `make DEBUG_HOOKS=1`): device int64 [blocks][8] that
// the epilogue of the fused launches over n_rows rows stamps, or NULL
extern "C" void pmc_debug_set_epilogue_stamps(long long* dev, int64_t n_rows) { g_epilogue_stamps = dev; g_epilogue_stamps_n = n_rows; }
#endif

extern "C" int pmc_stream_synchronize(void* stream) {
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return pmc_fail_hip(e, "hipStreamSynchronize");
    return 0;
}

// HIP events for live kernel timing from the host language (bench.py records the flow-inverse
// launch of every timed step through pmc_step_t.ev_inv0 / ev_inv1).
extern "C" void* pmc_event_create(void) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDefault) != hipSuccess) { pmc_fail("hipEventCreate failed"); return nullptr; }
    return (void*)e;
}
extern "C" int pmc_event_record(void* ev, void* stream) {
    hipError_t e = hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
    return e == hipSuccess ? 0 : pmc_fail_hip(e, "hipEventRecord");
}
// spin on a completion word the kernels store to pinned host memory (pmc_done_t)
extern "C" int pmc_wait_flag(const int64_t* flag, int64_t value, double timeout_s) {
    if (!flag) return pmc_fail("pmc_wait_flag: null flag");
    const volatile int64_t* f = flag;
    for (long it = 0;; ++it) {
        for (int k = 0; k < 2048; ++k) {
            if (*f == value) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return 0; }
            __builtin_ia32_pause();
        }
        if ((it & 63) == 63) {
            static thread_local struct timespec t0;
            struct timespec t;
            clock_gettime(CLOCK_MONOTONIC, &t);
            if (it == 63) t0 = t;
            if (timeout_s > 0.0 && (double)(t.tv_sec - t0.tv_sec) + 1e-9 * (double)(t.tv_nsec - t0.tv_nsec) > timeout_s)
                return pmc_fail("pmc_wait_flag: timed out waiting for the device");
        }
    }
}

extern "C" int pmc_event_synchronize(void* ev) {
    hipError_t e = hipEventSynchronize((hipEvent_t)ev);
    return e == hipSuccess ? 0 : pmc_fail_hip(e, "hipEventSynchronize");
}
extern "C" float pmc_event_elapsed_ms(void* a, void* b) {
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b) != hipSuccess) return -1.0f;
    return ms;
}
extern "C" void pmc_event_destroy(void* ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }


// ------------------------------------------------------------------------------------------------------------------
// The pipelined step of a walker set stepped as K row ranges (lanes), as ONE object behind the C ABI: the host language
// is left with the black boxes.  Per step (pocomc/mcmc.py:74-156) the driver thread does
//     pmc_pipeline_next(p, -1, ...)            wait for lane 0's x'
//     for k in lanes:  likelihood(lane k);  pmc_pipeline_next(p, k, ...)
// and every call of pmc_pipeline_next enqueues what the device can do next and returns when the host has something to
// do: the accept of lane k behind its logl', then either the wait for lane k+1's x', or -- behind the last lane -- the
// closing accept (total sums, sigma / mu update on the device, sums + completion word to the host), the pre-steps of
// step + 1 for all lanes, and the wait for the sums.  The same launches in the same order as mcmc.LanedEngine's
// step_pipelined (Python, round 2): results are bit-identical; the ~60 us of interpreter time per step are not.
struct pmc_pipeline {
    int n_lanes;
    const pmc_step_t* lanes[8];
    pmc_rng_t rng[8];
    int64_t step;                 // the step whose pre-steps are in flight
    void* stream;
    void* prefetcher;
    void* comm;                   // pmc_comm of a sharded walker set (the ranks' sums are exchanged behind the last accept), or NULL
    double timeout;
    double t_wait_x, t_wait_sums, t_enq_accept, t_enq_pre;
    int64_t n_steps;
};

static double now_s() {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

extern "C" void* pmc_pipeline_create(const pmc_step_t* const* lanes, int32_t n_lanes, uint64_t seed,
                                     const uint64_t* offsets, void* prefetcher, double wait_timeout_s, void* stream) {
    if (!lanes || n_lanes < 1 || n_lanes > 8 || !offsets) { pmc_fail("pmc_pipeline_create: 1..8 lanes"); return nullptr; }
    for (int k = 0; k < n_lanes; ++k) {
        const pmc_step_t* s = lanes[k];
        if (!s || !s->host_direct || !s->h_done || !s->done_ticket || !s->adapt_state || s->p_xT) {
            pmc_fail("pmc_pipeline_create: every lane needs host_direct buffers, completion words and adapt_state");
            return nullptr;
        }
        if (s->adapt_state != lanes[0]->adapt_state) { pmc_fail("pmc_pipeline_create: one adapt_state for all lanes"); return nullptr; }
    }
    pmc_pipeline* p = new pmc_pipeline();
    p->n_lanes = n_lanes;
    for (int k = 0; k < n_lanes; ++k) {
        p->lanes[k] = lanes[k];
        p->rng[k] = pmc_rng_t{nullptr, nullptr, nullptr, seed, 0, offsets[k]};
    }
    p->step = 0;
    p->stream = stream;
    p->prefetcher = prefetcher;
    p->comm = nullptr;
    p->timeout = wait_timeout_s;
    p->t_wait_x = p->t_wait_sums = p->t_enq_accept = p->t_enq_pre = 0.0;
    p->n_steps = 0;
    return p;
}

extern "C" void pmc_pipeline_destroy(void* pp) { delete (pmc_pipeline*)pp; }

// a sharded walker set (one process per GPU): behind the last lane's accept the ranks' sums are added by the library's own
// all-reduce (pmc_comm_adapt_update: IPC mailboxes, sums in rank order) before the adaptation -- the same pipeline as a
// single rank's, one more dependent launch
extern "C" int pmc_pipeline_set_comm(void* pp, void* comm) {
    pmc_pipeline* p = (pmc_pipeline*)pp;
    if (!p) return pmc_fail("pmc_pipeline_set_comm: null pipeline");
    p->comm = comm;
    return 0;
}

static int pipeline_enqueue_pre(pmc_pipeline* p, int64_t step, double nu) {
    for (int k = 0; k < p->n_lanes; ++k) {
        pmc_step_t s = *p->lanes[k];
        s.adapt_mode = 1;                                   // (pre: any non-zero mode = read sigma, cn_a, mu from adapt_state)
        p->rng[k].step = (uint64_t)step;
        const int rc = pmc_step_pre(&s, &p->rng[k], nu, 0.0, 0.0, p->stream);
        if (rc) return rc;
        if (p->prefetcher)                                  // helper threads read x' once as soon as its completion word shows up
            (void)pmc_prefetcher_submit(p->prefetcher, s.h_done, step + 1, s.h_x, (int64_t)s.n * s.D * 8, p->timeout > 0 ? p->timeout : 1.0);
    }
    return 0;
}

extern "C" int pmc_pipeline_start(void* pp, double nu, int64_t first_step) {
    pmc_pipeline* p = (pmc_pipeline*)pp;
    if (!p) return pmc_fail("pmc_pipeline_start: null pipeline");
    p->step = first_step;
    return pipeline_enqueue_pre(p, first_step, nu);
}

extern "C" int pmc_pipeline_next(void* pp, int32_t lane_done, double beta, double nu, int32_t adapt_mode, double c_sigma,
                                 double c_mu, double cap, double n_total, int32_t more) {
    pmc_pipeline* p = (pmc_pipeline*)pp;
    if (!p || lane_done < -1 || lane_done >= p->n_lanes) return pmc_fail("pmc_pipeline_next: bad argument");
    const int last = p->n_lanes - 1;
    double t0 = now_s();
    if (lane_done >= 0) {
        pmc_step_t s = *p->lanes[lane_done];
        s.adapt_n_other = 0;
        s.adapt_mode = 0;
        if (lane_done == last && !p->comm) {
            // the last range's accept closes the set: total sums, sigma / mu update, sums + completion word to the host
            s.adapt_mode = adapt_mode; s.adapt_c_sigma = c_sigma; s.adapt_c_mu = c_mu; s.adapt_cap = cap;
            s.adapt_n_total = n_total;
            for (int k = 0; k < last; ++k) s.adapt_other[k] = p->lanes[k]->sums;
            s.adapt_n_other = last;
        }
        const int rc = pmc_step_post(&s, &p->rng[lane_done], beta, nu, 0, (lane_done == last && !p->comm) ? 1 : 0, p->stream);
        if (rc) return rc;
        if (lane_done == last && p->comm) {
            // sharded: this rank's lanes + the other ranks' totals (rank order), the adaptation, sums + completion word
            const double* parts[8];
            for (int k = 0; k <= last; ++k) parts[k] = p->lanes[k]->sums;
            const pmc_step_t* sl = p->lanes[last];
            pmc_done_t dn{sl->h_done + 1, (int64_t)p->step + 1, nullptr};
            const int rc2 = pmc_comm_adapt_update(p->comm, parts, last + 1, sl->D, nullptr, sl->h_sums, sl->adapt_state, adapt_mode,
                                                  c_sigma, c_mu, cap, n_total, &dn, p->timeout, p->stream);
            if (rc2) return rc2;
        }
        const double t1 = now_s();
        p->t_enq_accept += t1 - t0;
        t0 = t1;
    }
    if (lane_done < last) {
        // the next lane's x'
        const pmc_step_t* nx = p->lanes[lane_done + 1];
        const int rc = pmc_wait_flag(nx->h_done, p->step + 1, p->timeout);
        p->t_wait_x += now_s() - t0;
        return rc;
    }
    if (more) {
        const int rc = pipeline_enqueue_pre(p, p->step + 1, nu);
        if (rc) return rc;
        const double t1 = now_s();
        p->t_enq_pre += t1 - t0;
        t0 = t1;
    }
    const int rc = pmc_wait_flag(p->lanes[last]->h_done + 1, p->step + 1, p->timeout);
    p->t_wait_sums += now_s() - t0;
    p->step += 1;
    p->n_steps += 1;
    return rc;
}

// out[0..5] <- { seconds waiting for x', for the sums, enqueuing accepts, enqueuing pre-steps, steps, 0 } since the last reset
extern "C" int pmc_pipeline_stats(void* pp, double* out, int32_t reset) {
    pmc_pipeline* p = (pmc_pipeline*)pp;
    if (!p || !out) return pmc_fail("pmc_pipeline_stats: null argument");
    out[0] = p->t_wait_x; out[1] = p->t_wait_sums; out[2] = p->t_enq_accept; out[3] = p->t_enq_pre;
    out[4] = (double)p->n_steps; out[5] = 0.0;
    if (reset) { p->t_wait_x = p->t_wait_sums = p->t_enq_accept = p->t_enq_pre = 0.0; p->n_steps = 0; }
    return 0;
}

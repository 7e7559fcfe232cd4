// Internal helpers shared by the .hip translation units.
#ifndef PMC_INTERNAL_H
#define PMC_INTERNAL_H

#include <hip/hip_runtime.h>
#include "../../include/pocomc_amd.h"

// A/B switches of measurement builds (make DEBUG_HOOKS=1): the product library never reads the environment.
#ifdef PMC_DEBUG_HOOKS
#include <stdlib.h>
static inline int pmc_env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
#define pmc_env_int(name, dflt) (dflt)
#endif

int pmc_fail(const char* msg);
int pmc_fail_hip(hipError_t e, const char* what);
int pmc_check_launch(const char* what);

int pmc_launch_forward_wg(const pmc_maf_t* m, const float* x, float* z, float* ladj, float* log_prob, int64_t n,
                          hipStream_t stream, const int64_t* idx = nullptr);
int pmc_launch_inverse_dpass_wg(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n, hipStream_t stream);
int pmc_launch_inverse_tri_nsf(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n, hipStream_t stream);
int pmc_launch_inverse_nsf2(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n, hipStream_t stream);   // -1: not covered
int pmc_launch_inverse_tri4(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n, hipStream_t stream, int variant = -1);   // variant: -1 by size, 0 solo, 1 duo
bool pmc_tri6_preferred(const pmc_maf_t* m);     // AUTO takes the lane-per-walker sweep for this flow (maf_inverse_tri6.hip)
int pmc_launch_propose_inverse_tri4(int kind, const float* cur32, const double* mu, const double* inv_cov,
                                    const double* chol, double nu, double sigma, double cn_a, const pmc_rng_t* rng,
                                    double* prop64, double* quad, double* quad_prop, const pmc_maf_t* m, float* x,
                                    float* ladj, int64_t n, hipStream_t stream, const double* adapt = nullptr,
                                    const struct ScalerEpi* epi = nullptr, int* epi_done = nullptr);   // (scaler_body.h)
int pmc_launch_propose_mfma(int kind, const float* cur32, const double* cur64, const double* mu,
                            const double* inv_cov, const double* chol, double nu, double sigma, double cn_a,
                            const pmc_rng_t* rng, double* prop64, float* prop32, double* quad, double* quad_prop,
                            int64_t n, int32_t D, hipStream_t stream, const double* adapt = nullptr);
int pmc_launch_clip_adamw(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                          double beta1, double beta2, double eps, double wd, double max_norm, int64_t step,
                          float* sq_scratch, hipStream_t st);
// pmc_scaler_inverse_prior with the step's extras: the number of rows that do not reach the likelihood (non-finite x' or
// logp') into *bad_flag in front of the completion word (bad_count: device word, zero between launches), and the walkers'
// current x (device f64 [n][D]) for those rows in the HOST copy of x' (pmc_step_t.fill_rejected)
struct pmc_scaler_extra { unsigned* bad_count; long long* bad_flag; const double* fill_x; };
int pmc_scaler_inverse_prior_ex(const pmc_scaler_t* s, const pmc_prior_t* prior, const float* u_in, const double* u_in64,
                                double* u_out, double* x, double* x_colmajor, double* logdetj, int32_t* finite, double* logp,
                                int32_t* finite_copy, double* logp_copy, const pmc_done_t* done, int64_t n, void* stream,
                                const pmc_scaler_extra* extra);
// pmc_propose with sigma / cn_a / mu taken from pmc_step_t.adapt_state (device) when adapt != NULL
int pmc_propose_adapt(int kind, const float* cur32, const double* cur64, const double* mu, const double* inv_cov,
                      const double* chol, double nu, double sigma, double cn_a, const pmc_rng_t* rng, double* prop64,
                      float* prop32, double* quad, double* quad_prop, int64_t n, int32_t D, void* stream,
                      const double* adapt);
// what the accept kernel's last block does with the sums (pmc_step_t.adapt_*)
struct pmc_adapt_args {
    double* state;
    int mode;
    double c_sigma, c_mu, cap, n_total;
    const double* other[7];     // sums of the other row ranges (pmc_step_t.adapt_other)
    int n_other;
};
int pmc_accept_adapt(int kind, int preconditioned, pmc_state_t* cur, const pmc_proposal_t* prop, double beta, double nu,
                     const pmc_rng_t* rng, double* alpha_out, int32_t* accept_out, double* sums, double* sums_copy,
                     const pmc_done_t* done, void* workspace, int64_t n, int32_t D, void* stream,
                     const pmc_adapt_args* adapt);

#endif

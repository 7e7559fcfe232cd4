// Proposal kernel on the f64 matrix cores (pocomc/mcmc.py:77-85, :251-253): propose_body.h, one wavefront per
// 16 walkers.
#include "propose_body.h"

template <int M>
__global__ __launch_bounds__(64) void propose_mfma_kernel(
    int kind, const float* __restrict__ cur32, const double* __restrict__ cur64,
    const double* __restrict__ mu, const double* __restrict__ inv_cov, const double* __restrict__ chol,
    double nu, double sigma, double cn_a, pmc_rng_t rng, double* __restrict__ prop64,
    float* __restrict__ prop32, double* __restrict__ quad, double* __restrict__ quad_prop,
    int64_t n, int D, const double* __restrict__ adapt) {
    if (adapt) { sigma = adapt[0]; cn_a = adapt[1]; mu = adapt + 2; }   // pmc_step_t.adapt_state
    propose_body<M>(kind, cur32, cur64, mu, inv_cov, chol, nu, sigma, cn_a, rng, prop64, prop32, quad, quad_prop, n, D,
                    nullptr, nullptr);
}

template <int M>
static int launch(int kind, const float* cur32, const double* cur64, const double* mu, const double* inv_cov,
                  const double* chol, double nu, double sigma, double cn_a, const pmc_rng_t* rng, double* prop64,
                  float* prop32, double* quad, double* quad_prop, int64_t n, int32_t D, hipStream_t stream,
                  const double* adapt) {
    hipLaunchKernelGGL(propose_mfma_kernel<M>, dim3((unsigned)((n + 15) / 16)), dim3(64), 0, stream, kind, cur32,
                       cur64, mu, inv_cov, chol, nu, sigma, cn_a, *rng, prop64, prop32, quad, quad_prop, n, (int)D, adapt);
    return pmc_check_launch("propose_mfma_kernel");
}

// returns -1 when D is too large for the register-resident kernel (caller falls back)
int pmc_launch_propose_mfma(int kind, const float* cur32, const double* cur64, const double* mu,
                            const double* inv_cov, const double* chol, double nu, double sigma, double cn_a,
                            const pmc_rng_t* rng, double* prop64, float* prop32, double* quad, double* quad_prop,
                            int64_t n, int32_t D, hipStream_t stream, const double* adapt) {
#define GO(MM) return launch<MM>(kind, cur32, cur64, mu, inv_cov, chol, nu, sigma, cn_a, rng, prop64, prop32, quad, \
                                 quad_prop, n, D, stream, adapt)
    if (D <= 16) GO(4);
    if (D <= 32) GO(8);
    if (D <= 64) GO(16);
    if (D <= 128) GO(32);
#undef GO
    return -1;
}

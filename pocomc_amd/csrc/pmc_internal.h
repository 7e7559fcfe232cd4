// Internal helpers shared by the .hip translation units.
#ifndef PMC_INTERNAL_H
#define PMC_INTERNAL_H

#include <hip/hip_runtime.h>
#include "../../include/pocomc_amd.h"

int pmc_fail(const char* msg);
int pmc_fail_hip(hipError_t e, const char* what);
int pmc_check_launch(const char* what);
int pmc_launch_inverse_tri2(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n, size_t lds,
                            hipStream_t stream);

int pmc_launch_forward_wg(const pmc_maf_t* m, const float* x, float* z, float* ladj, float* log_prob, int64_t n,
                          hipStream_t stream, const int64_t* idx = nullptr);
int pmc_launch_inverse_dpass_wg(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n, hipStream_t stream);
int pmc_launch_inverse_tri_nsf(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n, hipStream_t stream);
int pmc_launch_inverse_tri4(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n, hipStream_t stream, int variant = -1);   // variant: -1 by size, 0 solo, 1 duo
int pmc_launch_propose_inverse_tri4(int kind, const float* cur32, const double* mu, const double* inv_cov,
                                    const double* chol, double nu, double sigma, double cn_a, const pmc_rng_t* rng,
                                    double* prop64, double* quad, double* quad_prop, const pmc_maf_t* m, float* x,
                                    float* ladj, int64_t n, hipStream_t stream);
int pmc_launch_inverse_tri3(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n, hipStream_t stream);
int pmc_launch_propose_mfma(int kind, const float* cur32, const double* cur64, const double* mu,
                            const double* inv_cov, const double* chol, double nu, double sigma, double cn_a,
                            const pmc_rng_t* rng, double* prop64, float* prop32, double* quad, double* quad_prop,
                            int64_t n, int32_t D, hipStream_t stream);

#endif

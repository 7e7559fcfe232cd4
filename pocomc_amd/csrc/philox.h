// Counter-based RNG for the throughput path (Philox4x32-10, Salmon et al. 2011).
// The reference draws from numpy's global MT19937 stream (pocomc/mcmc.py:80,85,137);
// a sequential generator cannot be reproduced by a parallel kernel, so parity tests
// replay recorded variates and throughput runs use this generator: every variate is a
// pure function of (seed, step, global particle index, slot), independent of launch
// geometry and of the number of GPUs the population is sharded over.
#ifndef PMC_PHILOX_H
#define PMC_PHILOX_H

#include <hip/hip_runtime.h>
#include <stdint.h>

// A variate must not depend on the translation unit that draws it (pmc_rng_fill, the proposal kernel, the proposal as
// prologue of the flow-inverse kernels all promise the same bits): no implicit contraction in this header and after it.
#pragma clang fp contract(off)

struct Philox {
    uint32_t key[2];
    uint32_t ctr[4];

    __device__ __forceinline__ Philox(uint64_t seed, uint64_t step, uint64_t particle, uint32_t stream) {
        key[0] = (uint32_t)seed;
        key[1] = (uint32_t)(seed >> 32);
        ctr[0] = 0;                                  // draw counter within (step, particle, stream)
        ctr[1] = (uint32_t)particle;
        ctr[2] = (uint32_t)(particle >> 32) ^ (stream << 24);
        ctr[3] = (uint32_t)step ^ (uint32_t)(step >> 32) * 0x9E3779B9u;
    }

    __device__ __forceinline__ void next4(uint32_t out[4]) {
        uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
        uint32_t k0 = key[0], k1 = key[1];
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
            const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
            const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
            c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
        ++ctr[0];
    }

    // two uniforms in (0,1) with 53 random bits each
    __device__ __forceinline__ void uniform2(double& a, double& b) {
        uint32_t r[4];
        next4(r);
        const uint64_t x = ((uint64_t)r[0] << 32) | r[1], y = ((uint64_t)r[2] << 32) | r[3];
        a = ((double)(x >> 11) + 0.5) * (1.0 / 9007199254740992.0);
        b = ((double)(y >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    }

    // two independent N(0,1) (Box-Muller)
    __device__ __forceinline__ void normal2(double& a, double& b) {
        double u1, u2;
        uniform2(u1, u2);
        const double r = sqrt(-2.0 * log(u1));
        double s, c;
        sincospi(2.0 * u2, &s, &c);
        a = r * c;
        b = r * s;
    }

    // standard gamma, Marsaglia & Tsang (2000); shape > 0
    __device__ __forceinline__ double std_gamma(double shape) {
        double boost = 1.0;
        if (shape < 1.0) {
            double u, dummy;
            uniform2(u, dummy);
            boost = pow(u, 1.0 / shape);
            shape += 1.0;
        }
        const double d = shape - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
        for (int it = 0; it < 256; ++it) {
            double x, x2, u, u2;
            normal2(x, x2);
            uniform2(u, u2);
            // two candidates per Philox round trip
            for (int k = 0; k < 2; ++k) {
                const double xx = k ? x2 : x, uu = k ? u2 : u;
                const double t = 1.0 + c * xx;
                if (t > 0.0) {
                    const double v = t * t * t;
                    if (log(uu) < 0.5 * xx * xx + d - d * v + d * log(v)) return boost * d * v;
                }
            }
        }
        return boost * d;   // unreachable in practice (acceptance > 95% per candidate)
    }
};

#endif

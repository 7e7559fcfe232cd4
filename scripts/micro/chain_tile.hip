// Cycle counts of the pieces of the register chain of the triangular MAF inverse (csrc/maf_chain_rot.h), one wavefront
// alone on its SIMD (measurement only):
//   0: the four degree groups of a pattern-15 tile exactly as the sweeps compile them (chain_group_rot<15, 0, 4, 4, 1>)
//   1: one hop  MFMA -> read -> add, add, max  (dependent)
//   2: the same hop with one independent MFMA behind the critical one
//   3: the univariate map alone (dependent on itself)
//   4: 16 dependent v_add_f32
//   5: hop with the partial sum as the MFMA's accumulator input (one add less)
//   6: 16 dependent LDS round trips (ds_write_b32 -> ds_read_b32 of the same word)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../pocomc_amd/csrc -I../../include chain_tile.hip -o chain_tile
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "maf_chain_rot.h"

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, long long* cyc, int n, float seed) {
    __shared__ __attribute__((aligned(16))) float smem[4096];
    const int lane = threadIdx.x, q = lane >> 4, p = lane & 15;
    float* H0 = smem; float* H1 = smem + 1024; float* H2 = smem + 2048; float* X = smem + 2560; float* S = smem + 3584;
    for (int e = lane; e < 4096; e += 64) smem[e] = 0.0f;
    __syncthreads();
    float acc = 0.0f;
    long long t0 = 0, t1 = 0;
    if constexpr (MODE == 0) {
        ChainRot<4> s;
        for (int j = 0; j < 4; ++j) {
            s.p1[j] = 0.1f * j; s.p2[j] = -0.05f * j; s.a0[j] = 0.3f + 0.01f * lane;
            for (int i = 0; i < 4; ++i) s.w0r[i][j] = 0.02f * (i - j) * seed;
            s.po[j] = make_float2(0.1f, 0.2f * seed); s.yv[j] = 0.5f + j; s.g[j] = 4 + j;
        }
        s.wt1 = make_float4(0.01f * lane * seed, -0.02f * seed, 0.015f, 0.005f);
        s.wt2 = make_float4(-0.01f * lane * seed, 0.02f, -0.015f * seed, 0.004f);
        s.wo[0] = make_float4(0.01f, 0.02f * seed, 0.03f, 0.04f); s.wo[1] = make_float4(-0.01f, 0.02f, -0.03f * seed, 0.04f);
        float ladj = 0.0f;
        t0 = clock64();
        for (int it = 0; it < n; ++it) {
            s.acc1 = f32x4{0.f, 0.f, 0.f, 0.f}; s.acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
            s.outR[0] = f32x4{0.f, 0.f, 0.f, 0.f}; s.outR[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            chain_tile_begin(s, X, S, 32, q, p, lane);
            chain_group_rot<15, 0, 4, 4, 1>(s, H0, H1, X, it & 3, 32, 4, q, p, ladj, H2);
            chain_flush(s, ladj);
            s.a0[0] = s.a0[3] * 0.5f + s.pend_x;          // the next tile's first quad waits for this tile's last x
            s.a0[1] = 0.2f; s.a0[2] = 0.1f; s.a0[3] = 0.05f;
        }
        t1 = clock64();
        acc = ladj + s.a0[0];
    } else if constexpr (MODE == 1 || MODE == 2 || MODE == 5) {
        float w = 0.01f * seed + lane * 1e-4f, w2 = 0.02f * seed, h = 1.0f + lane * 1e-3f, pp = 0.1f * seed;
        f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
        t0 = clock64();
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (MODE == 5) {
                    c = MFMA(w, h, (f32x4{pp, pp, pp, pp}));
                    CHAIN_FENCE();
                    h = fmaxf(c[0] + h, 0.0f);
                } else {
                    c = MFMA(w, h, (f32x4{0.f, 0.f, 0.f, 0.f}));
                    CHAIN_FENCE();
                    if (MODE == 2) { d = MFMA(w2, h, d); CHAIN_FENCE(); }
                    h = fmaxf((c[0] + pp) + h, 0.0f);
                }
                CHAIN_FENCE();
            }
        }
        t1 = clock64();
        acc = h + d[0];
    } else if constexpr (MODE == 3) {
        float raw = 0.3f * seed, y = 0.7f, po = 0.1f * seed, x = 0.2f;
        t0 = clock64();
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float rr = x + po;
                const float ls = fast_ls(rr);
                x = (y - rr) * fast_exp_neg(ls);
                x = fmaf(0.01f, x, 0.3f);
                x = fmaxf(x, 0.0f);
            }
        }
        t1 = clock64();
        acc = x;
    } else if constexpr (MODE == 4) {
        float x = seed;
        t0 = clock64();
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { x = x + 1.0f; CHAIN_FENCE(); }
        }
        t1 = clock64();
        acc = x;
    } else {
        float x = seed;
        t0 = clock64();
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { S[lane] = x; WAVE_LDS_FENCE(); x = S[lane ^ 1] + 1.0f; WAVE_LDS_FENCE(); }
        }
        t1 = clock64();
        acc = x;
    }
    out[lane] = acc + smem[lane * 7];
    if (lane == 0) cyc[0] = t1 - t0;
}

int main() {
    float* o; long long* c; long long hc;
    hipMalloc(&o, 256); hipMalloc(&c, 8);
    const int n = 5000;
    const char* names[] = {"pattern-15 tile, 4 groups (chain_group_rot): cycles per tile", "hop (MFMA, read, add, add, max) x8: cycles per hop",
                           "hop + one independent MFMA x8: cycles per hop", "univariate map + fma + max x8: cycles each",
                           "16 dependent v_add_f32: cycles each", "hop with the partial as the accumulator input x8: cycles per hop",
                           "16 LDS write -> read round trips: cycles each"};
    const double div[] = {1, 8, 8, 8, 16, 8, 16};
#define RUN(M) for (int rep = 0; rep < 2; ++rep) { k<M><<<1, 64>>>(o, c, n, 1.0f); hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost); } \
    printf("%-70s %.1f\n", names[M], (double)hc / n / div[M]);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
    return 0;
}

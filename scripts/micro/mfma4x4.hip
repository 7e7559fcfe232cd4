// Semantics and latencies of v_mfma_f32_4x4x1_16b_f32 on gfx950 (scripts/micro: measurement only, not part of the library).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void sem_kernel(const float* a, const float* b, float* d0, float* d1) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    f32x4 r0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    f32x4 r1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 4, 5, 0);     // broadcast block 5's A to all 16 blocks
    for (int i = 0; i < 4; ++i) { d0[i * 64 + l] = r0[i]; d1[i * 64 + l] = r1[i]; }
}

template <int MODE>
__global__ void lat_kernel(float* out, long long* cyc, int n) {
    const int l = threadIdx.x;
    float a = 1.0f + l * 1e-3f, b = 1.0f - l * 1e-3f;
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) {            // dependent chain, same accumulator, 4x4x1
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 1, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 2, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 3, 0);
        } else if (MODE == 1) {     // 4 independent accumulators
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 4, 1, 0);
            c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 4, 2, 0);
            c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 4, 3, 0);
        } else if (MODE == 2) {     // dependent 16x16x4
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        } else if (MODE == 3) {     // layer hop: 4 dependent 4x4x1 -> VALU (add, max) -> feeds the next hop's B operand
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, c1[0], c0, 4, 1, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, c1[1], c0, 4, 2, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, c1[2], c0, 4, 3, 0);
            for (int r = 0; r < 4; ++r) c1[r] = fmaxf(c0[r] * 1e-3f + c1[r], 0.0f);
            b = c1[3];
            c0 = f32x4{0.f, 0.f, 0.f, 0.f};
        } else if (MODE == 4) {     // 2 interleaved dependent chains
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 4, 1, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 2, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 4, 3, 0);
        } else if (MODE == 6) {     // 16x16x4, 2 interleaved dependent chains
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        } else if (MODE == 7) {     // 16x16x4, 4 independent accumulators
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        } else if (MODE == 5) {     // VALU fma chain with 4 independent accumulators (16 fma)
            for (int r = 0; r < 4; ++r) { c0[r] = fmaf(a, b, c0[r]); c1[r] = fmaf(a, b, c1[r]); c2[r] = fmaf(a, b, c2[r]); c3[r] = fmaf(a, b, c3[r]); }
        }
    }
    long long t1 = clock64();
    out[l] = c0[0] + c1[1] + c2[2] + c3[3] + b;
    if (l == 0) cyc[0] = t1 - t0;
}

int main() {
    float ha[64], hb[64], h0[256], h1[256];
    for (int l = 0; l < 64; ++l) { ha[l] = 1 + l; hb[l] = 100 + l; }
    float *a, *b, *d0, *d1; long long* cyc;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d0, 1024); hipMalloc(&d1, 1024); hipMalloc(&cyc, 8);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    sem_kernel<<<1, 64>>>(a, b, d0, d1);
    hipMemcpy(h0, d0, 1024, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, 1024, hipMemcpyDeviceToHost);
    int bad0 = 0, bad1 = 0;
    for (int i = 0; i < 4; ++i) for (int L = 0; L < 64; ++L) {
        if (h0[i * 64 + L] != ha[4 * (L / 4) + i] * hb[L]) ++bad0;
        if (h1[i * 64 + L] != ha[4 * 5 + i] * hb[L]) ++bad1;
    }
    printf("semantics: D_i[L] = A[4*(L/4)+i] * B[L]: %s;  cbsz=4,abid=5: D_i[L] = A[20+i] * B[L]: %s\n", bad0 ? "NO" : "yes", bad1 ? "NO" : "yes");
    if (bad0 || bad1) { printf("d0[0..7] %g %g %g %g %g %g %g %g\n", h0[0], h0[1], h0[2], h0[3], h0[4], h0[5], h0[64], h0[65]); printf("d1[0..3] %g %g %g %g\n", h1[0], h1[1], h1[64], h1[65]); }
    const int n = 20000;
    long long hc;
    const char* names[] = {"4x4x1 dependent (same acc), per MFMA", "4x4x1 4 independent accs, per MFMA", "16x16x4 dependent, per MFMA",
                           "hop: 4 dep 4x4x1 + 4x(mul-add,max) + feed-back, per hop", "4x4x1 2 interleaved dep chains, per MFMA", "16 v_fma (4 indep x 4), per fma",
                           "16x16x4 2 interleaved dep chains, per MFMA", "16x16x4 4 independent accs, per MFMA"};
    float div[] = {4, 4, 4, 1, 4, 16, 4, 4};
#define RUN(M) lat_kernel<M><<<1, 64>>>(d0, cyc, n); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost); printf("%-62s %.1f ticks (clock64; s_memtime 100 MHz?)\n", names[M], (double)hc / n / div[M]);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    // calibrate clock64 tick vs wall: run long kernel
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); lat_kernel<1><<<1, 64>>>(d0, cyc, 2000000); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("calibration: %lld ticks in %.3f ms -> %.1f MHz tick; 8e6 MFMAs -> %.2f ns per MFMA\n", hc, ms, hc / ms / 1e3, ms * 1e6 / 8e6);
    return 0;
}

// Latency of a 4 x 4 block hop (h' = relu(a + W h + h)) done three ways on gfx950 (measurement only):
//   0: one v_mfma_f32_16x16x4_f32 with the quad's rows replicated (what the register chain of tri4 / tri5 does)
//   1: four v_fmac_f32 with DPP quad_perm broadcasts (a walker's quad in four adjacent lanes)
//   2: like 1, with a second quad's block interleaved (what "later quads" add to the stream)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K>
__device__ __forceinline__ float bq(float v) {        // value of lane (4 * (lane / 4) + K)
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), K * 0x55, 0xf, 0xf, true));
}

template <int MODE>
__global__ void hop_kernel(float* out, long long* cyc, int n) {
    const int l = threadIdx.x;
    float w0 = 0.01f + l * 1e-4f, w1 = -0.02f + l * 1e-4f, w2 = 0.015f, w3 = -0.005f, a = 0.1f;
    float h = 1.0f + l * 1e-3f, g = 0.5f;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) {
            c = f32x4{a, a, a, a};
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, h, c, 0, 0, 0);
            h = fmaxf(c[0] + h, 0.0f);
        } else if (MODE == 1) {
            float s = a;
            s = fmaf(w0, bq<0>(h), s); s = fmaf(w1, bq<1>(h), s); s = fmaf(w2, bq<2>(h), s); s = fmaf(w3, bq<3>(h), s);
            h = fmaxf(s + h, 0.0f);
        } else {
            float s = a, t = a;
            s = fmaf(w0, bq<0>(h), s); t = fmaf(w1, bq<0>(h), t);
            s = fmaf(w1, bq<1>(h), s); t = fmaf(w2, bq<1>(h), t);
            s = fmaf(w2, bq<2>(h), s); t = fmaf(w3, bq<2>(h), t);
            s = fmaf(w3, bq<3>(h), s); t = fmaf(w0, bq<3>(h), t);
            g += t;
            h = fmaxf(s + h, 0.0f);
        }
    }
    long long t1 = clock64();
    out[l] = h + g + c[1];
    if (l == 0) cyc[0] = t1 - t0;
}

__global__ void sem_kernel(float* out) {
    const int l = threadIdx.x;
    const float v = (float)l;
    out[l] = bq<0>(v) * 1000000.f + bq<1>(v) * 10000.f + bq<2>(v) * 100.f + bq<3>(v);
}

int main() {
    float* o; long long* c; long long hc;
    hipMalloc(&o, 256); hipMalloc(&c, 8);
    const int n = 20000;
    const char* names[] = {"hop via one 16x16x4 MFMA (replicated rows) + add + max", "hop via 4 v_fmac DPP quad_perm + add + max",
                           "the same with a second quad's block interleaved (8 fmac)"};
#define RUN(M) hop_kernel<M><<<1, 64>>>(o, c, n); hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost); printf("%-62s %.1f cycles\n", names[M], (double)hc / n);
    RUN(0) RUN(1) RUN(2)
    float ho[64];
    sem_kernel<<<1, 64>>>(o); hipMemcpy(ho, o, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) { const int b = l & ~3; if (ho[l] != b * 1000000.f + (b + 1) * 10000.f + (b + 2) * 100.f + (b + 3)) ++bad; }
    printf("quad_perm broadcast semantics: %s\n", bad ? "WRONG" : "ok");
    return 0;
}

// Latency / pacing of batched buffer loads of 512-byte blocks (the 16-bit helper fragments of maf_inverse_tri6.hip) for
// a lone wavefront per SIMD, with 1 or many workgroups reading the SAME image at the same time.
//   hipcc --offload-arch=gfx950 -O3 -o load_latency load_latency.hip && ./load_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// W = 8 / 16 bytes per lane; AW = waves of the workgroup that load (the others exit)
template <int NB, int W, int AW>
__global__ __launch_bounds__(256) void k(const unsigned char* img, int bytes, int rounds, long long* out, unsigned* sink) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv >= AW) { if (lane == 0) out[blockIdx.x * 4 + wv] = 0; return; }
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, bytes, 0x00020000);
    unsigned acc = 0;
    long long t0 = clock64();
    int off = (wv * 7919 * 1024) % (bytes - NB * 64 * W - 1024);
    off &= ~1023;
    for (int r = 0; r < rounds; ++r) {
        if constexpr (W == 8) {
            u32x2 v[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b64(rs, lane << 3, off + j * 512, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j) acc += v[j].x ^ v[j].y;
        } else {
            u32x4 v[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane << 4, off + j * 1024, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
        }
        off += NB * 64 * W;
        if (off > bytes - NB * 64 * W - 1024) off = 0;
    }
    long long t1 = clock64();
    if (lane == 0) out[blockIdx.x * 4 + wv] = t1 - t0;
    if (acc == 0x12345) sink[0] = acc;
}
int main() {
    const int bytes = 1500 * 1024;
    unsigned char* img; long long* out; unsigned* sink;
    hipMalloc(&img, bytes); hipMemset(img, 1, bytes); hipMalloc(&out, 1024 * 4 * 8); hipMalloc(&sink, 4);
    std::vector<long long> h(1024 * 4);
    for (int grid : {1, 157}) {
        const int rounds = 64;
#define RUN(NB, W, AW) { k<NB, W, AW><<<grid, 256>>>(img, bytes, rounds, out, sink); k<NB, W, AW><<<grid, 256>>>(img, bytes, rounds, out, sink); (void)hipDeviceSynchronize(); \
            (void)hipMemcpy(h.data(), out, grid * 32, hipMemcpyDeviceToHost); double s = 0; for (int i = 0; i < grid * 4; ++i) s += h[i]; \
            printf("grid %3d, %d loading waves, %2d B/lane, batch %2d: %8.0f cycles per batch, %.0f per load, %.1f B/clk/wave\n", grid, AW, W, NB, \
                   s / (grid * AW) / rounds, s / (grid * AW) / rounds / NB, 64.0 * W * NB / (s / (grid * AW) / rounds)); }
        RUN(1, 8, 1) RUN(8, 8, 1) RUN(24, 8, 1) RUN(8, 16, 1) RUN(24, 16, 1)
        RUN(8, 8, 2) RUN(24, 8, 2) RUN(8, 16, 2)
        RUN(8, 8, 4) RUN(24, 8, 4) RUN(8, 16, 4) RUN(24, 16, 4)
    }
    return 0;
}

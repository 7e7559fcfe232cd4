// Cycle counts of the pieces of the register chain of the triangular MAF inverse (csrc/maf_chain_rot.h), one wavefront
// alone on its SIMD (measurement only):
//   0: the four degree groups of a pattern-15 tile exactly as the sweeps compile them (chain_group_rot<15, 0, 4, 4, 1>)
//   1: one hop  MFMA -> read -> add, add, max  (dependent)
//   2: the same hop with one independent MFMA behind the critical one
//   3: the univariate map alone (dependent on itself)
//   4: 16 dependent v_add_f32
//   5: hop with the partial sum as the MFMA's accumulator input (one add less)
//   6: 16 dependent LDS round trips (ds_write_b32 -> ds_read_b32 of the same word)
//   7: the tile of the right-looking sweep (mode 0 + every group's share of the next tile + the tile-start reads)
//   8: mode 7 + the next tile's 16 fragment loads issued in the shadows of the hops
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../pocomc_amd/csrc -I../../include chain_tile.hip -o chain_tile
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "maf_chain_rot.h"
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

template <int MODE, int CABL = 1>
__global__ __launch_bounds__(64) void k(float* out, long long* cyc, int n, float seed, const float* wbuf) {
    __shared__ __attribute__((aligned(16))) float smem[4096];
    const int lane = threadIdx.x, q = lane >> 4, p = lane & 15;
    float* H0 = smem; float* H1 = smem + 1024; float* H2 = smem + 2048; float* X = smem + 2560; float* S = smem + 3584;
    for (int e = lane; e < 4096; e += 64) smem[e] = 0.0f;
    __syncthreads();
    float acc = 0.0f;
    long long t0 = 0, t1 = 0;
    if constexpr (MODE == 0 || MODE == 7 || MODE == 8) {
        ChainRot<4> s;
        for (int j = 0; j < 4; ++j) {
            s.p1[j] = 0.1f * j; s.p2[j] = -0.05f * j; s.a0[j] = 0.3f + 0.01f * lane;
            for (int i = 0; i < 4; ++i) s.w0r[i][j] = 0.02f * (i - j) * seed;
            s.po[j] = make_float2(0.1f, 0.2f * seed); s.yv[j] = 0.5f + j; s.g[j] = 4 + j;
        }
        s.wt1 = make_float4(0.01f * lane * seed, -0.02f * seed, 0.015f, 0.005f);
        s.wt2 = make_float4(-0.01f * lane * seed, 0.02f, -0.015f * seed, 0.004f);
        s.wo[0] = make_float4(0.01f, 0.02f * seed, 0.03f, 0.04f); s.wo[1] = make_float4(-0.01f, 0.02f, -0.03f * seed, 0.04f);
        s.wn1 = s.wt2; s.wn2 = s.wt1; s.woN[0] = s.wo[1]; s.woN[1] = s.wo[0];
        for (int j = 0; j < 4; ++j) { s.w0N[j] = make_float4(0.01f * j, 0.02f * seed, 0.03f, 0.01f * lane); s.a0N[j] = 0.0f; }
        s.accN1 = s.accN2 = s.outN[0] = s.outN[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)wbuf, 0, 1 << 20, 0x00020000);
        float4 ld[16];
        int it_ = 0;
        for (int j = 0; j < 16; ++j) ld[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        auto hook = [&](auto gi_, auto hop_) {
            constexpr int G = decltype(gi_)::value, HP = decltype(hop_)::value;
            if constexpr (MODE == 8) {
                constexpr int k = G * 3 + HP;           // 12 shadows, 16 loads: the first four shadows take two
                constexpr int a = k < 4 ? 2 * k : k + 4, b = k < 4 ? 2 * k + 2 : k + 5;
#pragma unroll
                for (int j = a; j < b; ++j) {
                    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane << 4, (it_ * 16 + j) * 1024, 0);
                    ld[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                }
            }
        };
        float ladj = 0.0f;
        t0 = clock64();
        for (int it = 0; it < n; ++it) {
            it_ = it & 31;
            if constexpr (MODE == 8) {                 // the fragments requested during the previous tile are this tile's
                s.wt1 = ld[0]; s.wt2 = ld[1]; s.wn1 = ld[2]; s.wn2 = ld[3]; s.wo[0] = ld[4]; s.wo[1] = ld[5]; s.woN[0] = ld[6]; s.woN[1] = ld[7];
                for (int j = 0; j < 4; ++j) { s.w0N[j] = ld[8 + j]; s.w0r[j][0] = ld[12 + j].x; s.w0r[j][1] = ld[12 + j].y; s.w0r[j][2] = ld[12 + j].z; s.w0r[j][3] = ld[12 + j].w; }
            }
            if constexpr (MODE >= 7) {                 // tile start of the right-looking sweep: staged partial + own share
                const float4 s0 = *reinterpret_cast<const float4*>(S + (lane << 2)), s1 = *reinterpret_cast<const float4*>(S + 256 + (lane << 2)), s2 = *reinterpret_cast<const float4*>(S + 512 + (lane << 2));
                s.a0[0] += s0.x + s.a0N[0]; s.a0[1] = s0.y + s.a0N[1]; s.a0[2] = s0.z + s.a0N[2]; s.a0[3] = s0.w + s.a0N[3];
                s.p1[0] = s1.x + s.accN1[0]; s.p1[1] = s1.y + s.accN1[1]; s.p1[2] = s1.z + s.accN1[2]; s.p1[3] = s1.w + s.accN1[3];
                s.p2[0] = s2.x + s.accN2[0]; s.p2[1] = s2.y + s.accN2[1]; s.p2[2] = s2.z + s.accN2[2]; s.p2[3] = s2.w + s.accN2[3];
                for (int j = 0; j < 4; ++j) {
                    const float2 so = *reinterpret_cast<const float2*>(S + 768 + (p << 4) + 2 * j);
                    s.po[j] = make_float2(so.x + s.outN[j >> 1][2 * (j & 1)], so.y + s.outN[j >> 1][2 * (j & 1) + 1]);
                    s.a0N[j] = 0.0f;
                }
                s.accN1 = s.accN2 = s.outN[0] = s.outN[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            s.acc1 = f32x4{0.f, 0.f, 0.f, 0.f}; s.acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
            s.outR[0] = f32x4{0.f, 0.f, 0.f, 0.f}; s.outR[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            chain_tile_begin(s, X, S, 32, q, p, lane);
            chain_group_rot<15, 0, 4, 4, CABL>(s, H0, H1, X, it & 3, 32, 4, q, p, ladj, H2, hook);
            chain_flush(s, ladj);
            s.a0[0] = s.a0[3] * 0.5f + s.pend_x;          // the next tile's first quad waits for this tile's last x
            s.a0[1] = 0.2f; s.a0[2] = 0.1f; s.a0[3] = 0.05f;
        }
        t1 = clock64();
        acc = ladj + s.a0[0] + ld[3].x;
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
                           "16 LDS write -> read round trips: cycles each",
                           "right-looking tile (own groups + share of the next tile, staged partials read): cycles per tile",
                           "the same with the next tile's 16 fragments requested in the hops' shadows: cycles per tile"};
    const double div[] = {1, 8, 8, 8, 16, 8, 16, 1, 1};
    float* wbuf; hipMalloc(&wbuf, 1 << 20); hipMemset(wbuf, 0, 1 << 20);
#define RUN(M) for (int rep = 0; rep < 2; ++rep) { k<M, (M >= 7 ? 3 : 1)><<<1, 64>>>(o, c, n, 1.0f, wbuf); hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost); } \
    printf("%-70s %.1f\n", names[M], (double)hc / n / div[M]);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
    return 0;
}

// Triangular-sweep MAF inverse, LANE-PER-WALKER chain (pocomc/flow.py:116-132, called every MCMC step: mcmc.py:88).
//
// What rocprof said about the register-chain kernels (profiles/r01_f_summary.txt, maf_inverse_tri5_kernel): 52 % of the
// issued MFMA work was padding -- the chain's 16x16x4 MFMAs carry the 4 rows of one quad replicated over the 16 tile
// rows -- and the sweep was a latency chain of ~2000 cycles per degree group for 16 walkers per wavefront.  Here
//
//   * the CHAIN wavefront owns up to 64 walkers, one per lane, and does everything that is sequential along the degree
//     groups with v_mfma_f32_4x4x1_16b_f32: D_i[lane] += A[4*abid + i] * B[lane] -- the B operand is the activation
//     register itself (no replication, no cross-lane move), the A operand one VGPR per 4 x 16 weight block
//     (lane l = W[row l & 3][k slot l >> 2], broadcast with cbsz = 4 / abid = k slot).  It multiplies only the
//     diagonal tile of the hidden layers, the layer-0 columns of the newest ranks and the output rows of its own ranks;
//   * three HELPER wavefronts (one per layer: hidden 1, hidden 2, layer 0 + output) multiply everything LEFT of the
//     diagonal tile with dense v_mfma_f32_16x16x4_f32 against tiles that are already final, for all the walker
//     subsets of the workgroup from one set of weight fragments, and stage the partial pre-activations in LDS;
//   * the wavefronts are coupled by monotonic LDS words ("tiles < v of layer l are final", "partials of tile v-1
//     are staged") instead of workgroup barriers: a wave's DS operations execute in order, so a data store followed
//     by the word's store needs no wait, and each helper starts on a tile as soon as ITS input layer is final --
//     h0 of a tile's last group is known three dependent hops before the tile ends.
//
// Wide flows (BASELINE configs[4]: D = 128, 8 transforms, H = 512 -> 33 hidden tiles, 16 output tiles; one workgroup of
// 16 walkers per CU, its three activation arrays take 101 KB of LDS).  In-kernel cycle stamps of workgroup 0
// (scripts/profile_tri6_config5.py) took this shape from 2.04 ms to 0.69 ms per sweep of <= 4096 walkers:
//   * a lone wavefront on its SIMD issues one instruction per ~4 cycles, so the helpers were bound by their instruction
//     count (45 per 16 x 16 block against four 32-cycle MFMAs), not by L2 or the matrix pipe: left_products() below;
//   * the hand-over words are addressed as LDS explicitly and read through readfirstlane -- as FLAT accesses every
//     poll waited for the weight fragments its wave had in flight, and a rank word read with a vector load made the
//     compiler wrap every buffer load that depends on it in a waterfall loop;
//   * the output wavefront keeps the sums of two output tiles across hidden tiles (a hidden tile adds one h2 tile
//     instead of a whole row) and starts a new output tile's row one hidden tile ahead of the chain; biases join the
//     sums when they are staged (as first value of an accumulator the bias load delays the first fragment request);
//   * flows of >= 20 hidden tiles run FIVE wavefronts: the layer-0 partials move to the fifth, and because two of five
//     share a SIMD the roles are assigned by SIMD id at run time (chain and output rows get a SIMD to themselves);
//   * the re-rank between transforms is done by the whole workgroup with the index loads batched (it was one wavefront
//     with two dependent global loads per element: 20 k cycles per transform).
// What bounds it now: below ~25 hidden tiles the chain wavefront (~4.2 k cycles per tile), above it the hidden-layer
// helpers, whose row of T blocks (128 cycles each at the matrix pipe's float32 rate) can only start one tile before
// it is needed.
//
// Same arithmetic as the other sweeps up to the order of additions (float32; parity vs oracle 1e-5 relative).
#include <stdlib.h>
#include <type_traits>
#include "maf_chain.h"
#include "propose_body.h"

typedef unsigned int u32x2_t6 __attribute__((ext_vector_type(2)));

#ifndef TRI6_ABL
#define TRI6_ABL 0                 // timing experiments only (scripts/abl_tri6.sh): results are wrong when != 0
#endif
#if TRI6_ABL != 0
// marks a library built with an ablation: pocomc_amd/_lib.py refuses it unless PMC_ALLOW_ABLATION is set
extern "C" int pmc_ablation_tri6(void) { return TRI6_ABL; }
#endif

namespace tri6 {

using ::bload4;                    // (maf_common.h)
__device__ __forceinline__ float2 bload2(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    const u32x2_t6 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}

template <int AB>
__device__ __forceinline__ f32x4 M4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, AB, 0);
}

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

// ---- operand types of the helper wavefronts' left-looking products ---------------------------------------------------
// HB = 0: float32 (v_mfma_f32_16x16x4_f32 x 4 per 16 x 16 block; fragments [lane][4] floats, activations float32 in LDS).
// HB = 1 / 2: bfloat16 / float16 operands, float32 accumulation (v_mfma_f32_16x16x16_bf16 / _f16, ONE per 16 x 16 block:
// ~16 cycles of matrix pipe against 128): the 16-bit image pmc_maf_pack_lane16 derives from the float32 fragments
// (pmc_maf_t.lane16: lane l = row l & 15, k = 4 (l >> 4) .. + 3, 8 bytes per lane and block) and activations the chain
// wavefront stores as 16-bit quads (lane (c, p) of a tile = units 4 c .. 4 c + 3 of walker p: half the LDS, which is what
// lets a second walker subset share the workgroup at D = 128).  The CHAIN stays float32 whatever HB: the diagonal tile of
// the hidden layers, the layer-0 columns of the newest ranks, the own output rows, the univariate map, the log-determinant;
// so do the layer-0 partials (x is float32).  Opt-in precision (Flow(precision="bf16")), stated tolerance in the tests.
typedef short s16x4_t6 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x4_t6 __attribute__((ext_vector_type(4)));
typedef __bf16 b16x2_t6 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2_t6 __attribute__((ext_vector_type(2)));
typedef float f32x2_t6 __attribute__((ext_vector_type(2)));
typedef __bf16 b16x8_t6 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x8_t6 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint2 bload2u(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    const u32x2_t6 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    return make_uint2(v.x, v.y);
}

__device__ __forceinline__ uint4 bload4u(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Fragment rows: a row of k tiles is k blocks of 1 KiB (float32), or ceil(k / 2) PAIRS of 1 KiB with 16-bit operands -- lane
// (c, i) of a pair holds W[i][4 c .. 4 c + 3] of tile 2 b in its first 8 bytes and of tile 2 b + 1 in the next 8, so that ONE
// 16-byte load per lane feeds two tiles' MFMAs.  (scripts/micro/load_latency.hip: with two or more wavefronts of a CU
// loading, a wavefront gets one buffer load back per ~90-110 cycles WHATEVER its width, 8 or 16 bytes per lane -- the
// helpers' 16-bit products are paced by the NUMBER of load instructions, not by bytes or by the matrix pipe.)
template <int HB> struct Ops;
template <> struct Ops<0> {
    using W = float4;
    static constexpr int BLK = 1024;                      // bytes of one 16 x 16 fragment block
    static constexpr int TF = 256;                        // floats of one tile of an activation array (per subset)
    static __device__ __forceinline__ int row_bytes(int tiles) { return tiles << 10; }
    static __device__ __forceinline__ int act_floats(int tiles) { return tiles << 8; }
    static __device__ __forceinline__ W zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ W loadw(__amdgpu_buffer_rsrc_t rs, int lane, int soff) { return bload4(rs, lane << 4, soff); }
    static __device__ __forceinline__ W loadw_tile(__amdgpu_buffer_rsrc_t rs, int lane, int rowbase, int t) { return bload4(rs, lane << 4, rowbase + (t << 10)); }
    static __device__ __forceinline__ W loadb(const float* H, int K, int lane) {
        return *reinterpret_cast<const float4*>(H + (K << 8) + (lane << 2));
    }
    static __device__ __forceinline__ void mma(f32x4& acc, f32x4& acd, const W& w, const W& b, int) {
        acc = MFMA(w.x, b.x, acc); acd = MFMA(w.y, b.y, acd);
        acc = MFMA(w.z, b.z, acc); acd = MFMA(w.w, b.w, acd);
    }
};
template <int HB> struct Ops {
    using W = uint2;                                      // a single tile's fragment / B operand (half of a pair's 16 bytes)
    static constexpr int TF = 128;                        // floats per tile of an activation array: pairs of tiles are 256
    static __device__ __forceinline__ int row_bytes(int tiles) { return ((tiles + 1) >> 1) << 10; }
    static __device__ __forceinline__ int act_floats(int tiles) { return ((tiles + 1) >> 1) << 8; }
    static __device__ __forceinline__ W zero() { return make_uint2(0u, 0u); }
    static __device__ __forceinline__ W loadw_tile(__amdgpu_buffer_rsrc_t rs, int lane, int rowbase, int t) {
        return bload2u(rs, (lane << 4) + ((t & 1) << 3), rowbase + ((t >> 1) << 10));
    }
    // activations: [pair of tiles][lane (c, p)][tile 2 b: units 4 c .. 4 c + 3 | tile 2 b + 1: the same] -- 16 bytes per lane
    static __device__ __forceinline__ W loadb(const float* H, int t, int lane) {
        return *reinterpret_cast<const uint2*>(H + ((t >> 1) << 8) + (lane << 2) + ((t & 1) << 1));
    }
    static __device__ __forceinline__ uint4 loadb_pair(const float* H, int pair, int lane) {
        return *reinterpret_cast<const uint4*>(H + (pair << 8) + (lane << 2));
    }
    static __device__ __forceinline__ f32x4 mfma16(const W& w, const W& b, f32x4 c) {
        if constexpr (HB == 1)
            return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*reinterpret_cast<const s16x4_t6*>(&w), *reinterpret_cast<const s16x4_t6*>(&b), c, 0, 0, 0);
        else
            return __builtin_amdgcn_mfma_f32_16x16x16f16(*reinterpret_cast<const h16x4_t6*>(&w), *reinterpret_cast<const h16x4_t6*>(&b), c, 0, 0, 0);
    }
    // a pair of tiles in one instruction: k = 8 c + j  <->  tile 2 b + (j >> 2), unit 4 c + (j & 3), the same in A and B
    static __device__ __forceinline__ f32x4 mfma32(const uint4& w, const uint4& b, f32x4 c) {
        if constexpr (HB == 1)
            return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const b16x8_t6*>(&w), *reinterpret_cast<const b16x8_t6*>(&b), c, 0, 0, 0);
        else
            return __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const h16x8_t6*>(&w), *reinterpret_cast<const h16x8_t6*>(&b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma(f32x4& acc, f32x4& acd, const W& w, const W& b, int j) {
        if (j & 1) acd = mfma16(w, b, acd); else acc = mfma16(w, b, acc);
    }
};
// the chain's store of one quad (units 4 c .. 4 c + 3 of its walker) of a hidden tile as 16-bit values: 8 bytes
template <int HB>
__device__ __forceinline__ void store_quad16(float* H, int T, int c, int p, const f32x4& v) {
    uint2 w;
    if constexpr ((TRI6_ABL & 2) != 0) {
        w.x = __float_as_uint(v[0]); w.y = __float_as_uint(v[2]);
    } else if constexpr (HB == 1) {
        const b16x2_t6 a = __builtin_convertvector(f32x2_t6{v[0], v[1]}, b16x2_t6), b = __builtin_convertvector(f32x2_t6{v[2], v[3]}, b16x2_t6);
        w.x = *reinterpret_cast<const unsigned*>(&a); w.y = *reinterpret_cast<const unsigned*>(&b);
    } else {
        const h16x2_t6 a = __builtin_convertvector(f32x2_t6{v[0], v[1]}, h16x2_t6), b = __builtin_convertvector(f32x2_t6{v[2], v[3]}, h16x2_t6);
        w.x = *reinterpret_cast<const unsigned*>(&a); w.y = *reinterpret_cast<const unsigned*>(&b);
    }
    *reinterpret_cast<uint2*>(H + ((T >> 1) << 8) + (((c << 4) + p) << 2) + ((T & 1) << 1)) = w;
}

// ---- LDS words ----------------------------------------------------------------------------------------------------
enum { F_H0 = 0, F_H1, F_H2, F_X, F_P0, F_P1, F_P2, F_P3, F_COUNT = 8 };

// The words are addressed as LDS (address space 3) explicitly: through a generic pointer they become FLAT accesses,
// which count on vmcnt as well -- every poll then waits for the weight fragments its wave has in flight (measured:
// 1300 cycles per tile on the chain wavefront, a full L2 miss per row on the helpers).
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ volatile lds_int* lds_word(const int* flags, int which) {
    return reinterpret_cast<volatile lds_int*>(static_cast<unsigned>(reinterpret_cast<uintptr_t>(flags + which)));   // (aperture: low 32 bits)
}
// (every lane reads the same word: said explicitly, the polls and the branches on them are scalar)
__device__ __forceinline__ int peek(const int* flags, int which) { return __builtin_amdgcn_readfirstlane(*lds_word(flags, which)); }
__device__ __forceinline__ void publish(int* flags, int which, int value) {
    asm volatile("" ::: "memory");                       // data stores stay before the word's store (DS ops are in order)
    *lds_word(flags, which) = value;
}
// NAP: the helper wavefronts' polls sleep between two looks (s_sleep 1 = 64 cycles).  With 16-bit operands a helper is done
// with its row thousands of cycles before the chain publishes the next tile; three wavefronts polling back to back put an
// LDS instruction in front of every DS operation of the chain (measured: chain tile 4.3 k -> 5.3 k cycles)
template <bool NAP = false>
__device__ __forceinline__ void wait_for(const int* flags, int which, int value) {
    while (peek(flags, which) < value) { if constexpr (NAP && !(TRI6_ABL & 8)) __builtin_amdgcn_s_sleep(1); }
    asm volatile("" ::: "memory");
}

constexpr int SPAD = 20;                                  // floats per walker row of a staging tile (16 + 4: conflict-free b128)

struct ChainState {
    f32x4 a0[4], a0n[4], acc1[4], acc2[4], o[2];
    float4 w1, w2, w0, w0n;                               // A operands: component = out quad
    float2 w3;                                            // A operands of the two group pairs' output rows
    float yv[4];
    int g[4];
};

// the chain's store of x_g: float32 over y_g IN PLACE (the array holds the transform's input by rank; a rank's y is read
// before its x is written, and what the layer-0 helper multiplies of the ranks not solved yet meets zero weights only: f0c),
// and, with 16-bit helpers, one element of the 16x16x16 B layout the layer-0 helper reads (lane ((g & 15) >> 2, p) of tile
// g >> 4, element g & 3)
template <int HB>
__device__ __forceinline__ void store_x(float* A, float* X16, int g, int p, float v) {
    A[lidx(g, p)] = v;
    if constexpr (HB != 0 && !(TRI6_ABL & 1)) {
        unsigned short h;
        if constexpr (HB == 1) { const __bf16 b = (__bf16)v; h = *reinterpret_cast<const unsigned short*>(&b); }
        else { const _Float16 b = (_Float16)v; h = *reinterpret_cast<const unsigned short*>(&b); }
        reinterpret_cast<unsigned short*>(X16)[((((g >> 5) << 6) + (((g & 15) >> 2) << 4) + p) << 3) + (((g >> 4) & 1) << 2) + (g & 3)] = h;
    }
}
// groups of a tile from its four quad degrees (rank words): s.g[i] = rank of group i, D = no such group
__device__ __forceinline__ void tile_groups(const int4& dg, int D, int (&g)[4]) {
    const bool ny = dg.y != dg.x, nz = dg.z != dg.y, nw = dg.w != dg.z;
    g[0] = dg.x;
    g[1] = ny ? dg.y : (nz ? dg.z : (nw ? dg.w : D));
    g[2] = ny ? (nz ? dg.z : (nw ? dg.w : D)) : ((nz && nw) ? dg.w : D);
    g[3] = (ny && nz && nw) ? dg.w : D;
}

// Speculative hand-over read: the word and the staged data are read in ONE LDS round trip (DS operations of a wave
// execute in order and the writer stored the data before the word, so data read after a word that already shows
// `value` are the staged ones); only if the word is behind does the wave poll and read again.
#if (TRI6_ABL & 32)
__device__ unsigned long long g_tri6_wait[2 * F_COUNT];      // measurement build: [word] cycles the chain of workgroup 0 waited, [F_COUNT + word] times
#endif
template <class LOAD>
__device__ __forceinline__ void take(const int* flags, int which, int value, LOAD&& load) {
    const int seen = peek(flags, which);
    load();
    asm volatile("" ::: "memory");
    if (seen < value) {
#if (TRI6_ABL & 32)
        const long long t0 = clock64();
#endif
        wait_for(flags, which, value);
#if (TRI6_ABL & 32)
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
            atomicAdd(&g_tri6_wait[which], (unsigned long long)(clock64() - t0));
            atomicAdd(&g_tri6_wait[F_COUNT + which], 1ull);
        }
#endif
        load();
    }
}

// acc/acd[sb] += sum_{K < Kn} frag[K] . Hin[sb][K] for a helper wavefront, every tile K < Kn final when `ready` returns
// (it is called between the first fragment loads and the first activation reads).
//
// What bounds a helper (profile of workgroup 0 at D = 128, 33 hidden tiles; builds without the weight loads, without
// the LDS reads and without the MFMAs all ran within 10 % of each other): a lone wavefront on its SIMD issues one
// instruction per ~4 cycles, so a tile's four 32-cycle MFMAs hide about 28 other instructions.  The first version of
// this loop spent 45 per tile (per-tile bounds tests, selects, address arithmetic): 180-250 cycles per tile.  Now the
// loop body is one fragment load, one LDS read per subset and the MFMAs: chunks of C tiles, addressed by immediate
// offsets from one base per chunk, three register sets in rotation (a copy of a set would wait for its loads) so that
// the fragments (L2) and activations (LDS) of chunks i + 1 and i + 2 are in flight while chunk i multiplies; only whole
// rotations of FULL chunks run in the loop (no bounds test inside; an exit between the stages would also make the wait
// counts of the merged paths conservative), the last tiles (fewer than three chunks, two of them already loaded)
// follow it behind one uniform branch each.  The empty asm and the scheduling barriers keep the compiler from moving a
// set's loads down to the stage that multiplies them.
// PRE: the caller requested the row's first two chunks into w0 / w1 already (first_chunks() below: the hidden-layer helpers
// do it for their NEXT row while they wait for the last input tile of the current one).
// ---- float32 fragments: chunks of C tiles, three register sets in rotation (see the comment above)
template <int C>
__device__ __forceinline__ void first_chunks32(float4 (&w0)[C], float4 (&w1)[C], __amdgpu_buffer_rsrc_t rs, int lane, int base, int Kn) {
#pragma unroll
    for (int j = 0; j < C; ++j) w0[j] = bload4(rs, lane << 4, base + j * 1024);
    const int so = base + ((C < Kn ? C : 0) << 10);
#pragma unroll
    for (int j = 0; j < C; ++j) w1[j] = bload4(rs, lane << 4, so + j * 1024);
}
template <int NS, int C, bool PRE, class READY>
__device__ __forceinline__ void left32_w(f32x4 (&acc)[NS], f32x4 (&acd)[NS], __amdgpu_buffer_rsrc_t rs, int base,
                                         const float* Hin, int szH, int Kn, int lane, float4 (&w0)[C], float4 (&w1)[C], READY&& ready) {
    if (Kn <= 0) return;
    using O = Ops<0>;
    float4 w2[C], b0[C][NS], b1[C][NS], b2[C][NS];
    // chunk at tile Kc (a chunk that starts beyond the row is not used: tile 0 instead, always inside the arrays)
    auto loadw = [&](float4 (&w)[C], int Kc) __attribute__((always_inline)) {
        const int so = base + ((Kc < Kn ? Kc : 0) << 10);
#pragma unroll
        for (int j = 0; j < C; ++j) w[j] = bload4(rs, lane << 4, so + j * 1024);
    };
    auto loadb = [&](float4 (&b)[C][NS], int Kc) __attribute__((always_inline)) {
        const int K0 = Kc < Kn ? Kc : 0;
#pragma unroll
        for (int j = 0; j < C; ++j)
#pragma unroll
            for (int sb = 0; sb < NS; ++sb) b[j][sb] = O::loadb(Hin + sb * szH, K0 + j, lane);
    };
    auto mma1 = [&](const float4& w, const float4 (&b)[NS], int j) __attribute__((always_inline)) {
#pragma unroll
        for (int sb = 0; sb < NS; ++sb) O::mma(acc[sb], acd[sb], w, b[sb], j);
    };
#define STAGE(LW, LB, MW, MB)                                                                                       \
        loadw(LW, K0 + 2 * C); loadb(LB, K0 + 2 * C);                                                               \
        asm volatile("" ::: "memory");                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        _Pragma("unroll") for (int j = 0; j < C; ++j) mma1(MW[j], MB[j], j);                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        K0 += C;
    if constexpr (!PRE) { loadw(w0, 0); loadw(w1, C); }
    asm volatile("" ::: "memory");
    ready();                                              // (the wait for the tiles' word, behind the first fragments' loads)
    loadb(b0, 0); loadb(b1, C);
    asm volatile("" ::: "memory");
    int K0 = 0;
    for (int it = (Kn / C) / 3; it > 0; --it) {
        STAGE(w2, b2, w0, b0)
        STAGE(w0, b0, w1, b1)
        STAGE(w1, b1, w2, b2)
    }
#undef STAGE
    if (K0 + 2 * C < Kn) {                                // the third chunk of the rest
        loadw(w2, K0 + 2 * C); loadb(b2, K0 + 2 * C);
        asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int j = 0; j < C; ++j) if (K0 + j < Kn) mma1(w0[j], b0[j], j);
#pragma unroll
    for (int j = 0; j < C; ++j) if (K0 + C + j < Kn) mma1(w1[j], b1[j], j);
#pragma unroll
    for (int j = 0; j < C; ++j) if (K0 + 2 * C + j < Kn) mma1(w2[j], b2[j], j);
}

// ---- 16-bit fragments: the same rotation over chunks of C PAIRS of tiles; tiles [K0, K1) of the row at `rowbase`
template <int C>
__device__ __forceinline__ void first_chunks16(uint4 (&w0)[C], uint4 (&w1)[C], __amdgpu_buffer_rsrc_t rs, int lane, int rowbase, int K1) {
    const int p1 = K1 >> 1;                               // (full pairs of the range [0, K1))
#pragma unroll
    for (int j = 0; j < C; ++j) w0[j] = bload4u(rs, lane << 4, rowbase + j * 1024);
    const int so = rowbase + ((C < p1 ? C : 0) << 10);
#pragma unroll
    for (int j = 0; j < C; ++j) w1[j] = bload4u(rs, lane << 4, so + j * 1024);
}
template <int HB, int NS, int C, bool PRE, class READY>
__device__ __forceinline__ void left16_w(f32x4 (&acc)[NS], f32x4 (&acd)[NS], __amdgpu_buffer_rsrc_t rs, int rowbase,
                                         const float* Hin, int szH, int K0, int K1, int lane, uint4 (&w0)[C], uint4 (&w1)[C],
                                         READY&& ready) {
    if (K1 <= K0) return;
    using O = Ops<HB>;
    // the range as  [a leading single tile]  full pairs [pf0, pf1)  [a trailing single tile]: a full pair is ONE 16-byte
    // fragment load, one 16-byte LDS read per subset and one v_mfma_f32_16x16x32 per subset -- no per-tile scalar work (a lone
    // wavefront issues an instruction per ~5 cycles and a 16-bit MFMA no longer hides any: the loop's instruction count is
    // its time)
    const bool lead = (K0 & 1) != 0;                      // tile K0 = second half of pair K0 >> 1
    const int pf0 = (K0 + 1) >> 1, pf1 = K1 >> 1;
    const bool trail = (K1 & 1) != 0 && !(lead && K1 - 1 == K0);      // tile K1 - 1 = first half of pair pf1
    uint2 wl = make_uint2(0u, 0u), wt = make_uint2(0u, 0u);
    if (lead) wl = O::loadw_tile(rs, lane, rowbase, K0);
    if (trail) wt = O::loadw_tile(rs, lane, rowbase, K1 - 1);
    const int np = pf1 > pf0 ? pf1 - pf0 : 0;
    uint4 w2[C], b0[C][NS], b1[C][NS], b2[C][NS];
    // chunk at pair pc (a chunk that starts beyond the range is not used: pair 0 of the row instead, always inside the image;
    // pairs of a chunk beyond pf1 are loaded and never multiplied)
    auto loadw = [&](uint4 (&w)[C], int pc) __attribute__((always_inline)) {
        const int so = rowbase + ((pc < pf1 ? pc : 0) << 10);
#pragma unroll
        for (int j = 0; j < C; ++j) w[j] = bload4u(rs, lane << 4, so + j * 1024);
    };
    auto loadb = [&](uint4 (&b)[C][NS], int pc) __attribute__((always_inline)) {
        const float* hk = Hin + ((pc < pf1 ? pc : 0) << 8);
#pragma unroll
        for (int j = 0; j < C; ++j)
#pragma unroll
            for (int sb = 0; sb < NS; ++sb) b[j][sb] = O::loadb_pair(hk + sb * szH, j, lane);
    };
    auto mma1 = [&](const uint4& w, const uint4 (&b)[NS], int j) __attribute__((always_inline)) {
#pragma unroll
        for (int sb = 0; sb < NS; ++sb) {
            if (j & 1) acd[sb] = O::mfma32(w, b[sb], acd[sb]); else acc[sb] = O::mfma32(w, b[sb], acc[sb]);
        }
    };
#define STAGE(LW, LB, MW, MB)                                                                                       \
        loadw(LW, P + 2 * C); loadb(LB, P + 2 * C);                                                                 \
        asm volatile("" ::: "memory");                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        _Pragma("unroll") for (int j = 0; j < C; ++j) mma1(MW[j], MB[j], j);                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        P += C;
    if constexpr (!PRE) { if (np > 0) { loadw(w0, pf0); loadw(w1, pf0 + C); } }
    asm volatile("" ::: "memory");
    ready();                                              // (the wait for the tiles' word, behind the first fragments' loads)
    if (np > 0) {
        loadb(b0, pf0); loadb(b1, pf0 + C);
        asm volatile("" ::: "memory");
        int P = pf0;
        for (int it = (np / C) / 3; it > 0; --it) {
            STAGE(w2, b2, w0, b0)
            STAGE(w0, b0, w1, b1)
            STAGE(w1, b1, w2, b2)
        }
        if (P + 2 * C < pf1) {                            // the third chunk of the rest
            loadw(w2, P + 2 * C); loadb(b2, P + 2 * C);
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int j = 0; j < C; ++j) if (P + j < pf1) mma1(w0[j], b0[j], j);
#pragma unroll
        for (int j = 0; j < C; ++j) if (P + C + j < pf1) mma1(w1[j], b1[j], j);
#pragma unroll
        for (int j = 0; j < C; ++j) if (P + 2 * C + j < pf1) mma1(w2[j], b2[j], j);
    }
#undef STAGE
    if (lead) {
#pragma unroll
        for (int sb = 0; sb < NS; ++sb) acc[sb] = O::mfma16(wl, O::loadb(Hin + sb * szH, K0, lane), acc[sb]);
    }
    if (trail) {
#pragma unroll
        for (int sb = 0; sb < NS; ++sb) acd[sb] = O::mfma16(wt, O::loadb(Hin + sb * szH, K1 - 1, lane), acd[sb]);
    }
}

// acc/acd[sb] += sum_{K in [K0, K1)} frag[row][K] . H[sb][K]   (rowbase: byte offset of the row's first block / pair; Hin: tile 0
// of subset 0; every tile of the range final when `ready` returns).  C: tiles (float32) / pairs (16-bit) per chunk.
template <int HB, int NS, int C, class READY>
__device__ __forceinline__ void left_products(f32x4 (&acc)[NS], f32x4 (&acd)[NS], __amdgpu_buffer_rsrc_t rs, int rowbase,
                                              const float* Hin, int szH, int K0, int K1, int lane, READY&& ready) {
    if constexpr (HB == 0) {
        float4 w0[C], w1[C];
        left32_w<NS, C, false>(acc, acd, rs, rowbase + (K0 << 10), Hin + (K0 << 8), szH, K1 - K0, lane, w0, w1, ready);
    } else {
        uint4 w0[C], w1[C];
        left16_w<HB, NS, C, false>(acc, acd, rs, rowbase, Hin, szH, K0, K1, lane, w0, w1, ready);
    }
}

// One degree group of the tile (quads c0..c1), then the next (compile-time recursion over the quad pattern).
// Order of the MFMAs: the matrix pipe of a lone wave executes in order, 15 cycles per 4x4x1 (scripts/micro/mfma4x4.hip),
// and the group's critical path is  h0 -> [own block of layer 1] -> h1 -> [own block of layer 2] -> h2 -> [own output
// rows] -> x -> [layer-0 column of the next quad].  The blocks that feed LATER quads are issued in the gaps the VALU
// epilogues of that path leave (about four MFMAs each); what does not fit follows at the end of the group.
template <int PAT, int I, int HB>
__device__ __forceinline__ void chain_group6(ChainState& s, float* H0, float* H1, float* H2, float* X, float* X16, int* flags,
                                             const float* SP1, const float* SP2, const float* SP3, int T, int D,
                                             int gen, int sub_off_sp, int sp3_rows, int p, bool writer, float& ladj) {
    constexpr int NG = pat_ngroups(PAT);
    if constexpr (I < NG) {
        constexpr int c0 = pat_start(PAT, I), c1 = pat_end(PAT, I);
        constexpr bool last = (I == NG - 1);
        constexpr int nx = c1 + 1;                        // first quad after this group
        const int g = s.g[I];
        const bool live = g < D;
        const int hw = (T << 8) + (p << 2);               // + (unit << 6) + quad
        f32x4 h0[4], h1[4], h2[4];
        // blocks of layer L (1 / 2) from this group's quads into out quads [A0, A1)
#define BLOCKS(ACC, W, HV, A0, A1)                                                                                  \
        sfor<c0, c1 + 1>([&](auto bb) __attribute__((always_inline)) {                                              \
            constexpr int b = decltype(bb)::value;                                                                  \
            sfor<0, 4>([&](auto kk) __attribute__((always_inline)) {                                                \
                constexpr int k = decltype(kk)::value;                                                              \
                sfor<(A0), (A1)>([&](auto aa) __attribute__((always_inline)) {                                      \
                    constexpr int a = decltype(aa)::value;                                                          \
                    s.ACC[a] = M4<4 * b + k>(comp(s.W, a), HV[b][k], s.ACC[a]);                                     \
                });                                                                                                 \
            });                                                                                                     \
        });
        // ---- layer 0
        sfor<c0, c1 + 1>([&](auto cc) __attribute__((always_inline)) {
            constexpr int c = decltype(cc)::value;
            for (int r = 0; r < 4; ++r) h0[c][r] = fmaxf(s.a0[c][r], 0.0f);
            if (writer) {
                if constexpr (HB == 0) { for (int r = 0; r < 4; ++r) H0[hw + (r << 6) + c] = h0[c][r]; }
                else store_quad16<HB>(H0, T, c, p, h0[c]);
            }
        });
        if constexpr (last) publish(flags, F_H0, gen + T + 1);
        if constexpr (I == 0) {                           // partial pre-activations of layer 1 (tiles left of this one)
            float4 v[4];
            take(flags, F_P1, gen + T + 1, [&]() __attribute__((always_inline)) {
                for (int a = 0; a < 4; ++a) v[a] = *reinterpret_cast<const float4*>(SP1 + sub_off_sp + 4 * a);
            });
            for (int a = 0; a < 4; ++a) s.acc1[a] = f32x4{v[a].x, v[a].y, v[a].z, v[a].w};
        }
        BLOCKS(acc1, w1, h0, c0, c1 + 1)                  // critical: own quads
        if constexpr (nx < 4) { BLOCKS(acc1, w1, h0, nx, nx + 1) }          // gap: the next quad
        // ---- layer 1
        sfor<c0, c1 + 1>([&](auto cc) __attribute__((always_inline)) {
            constexpr int c = decltype(cc)::value;
            for (int r = 0; r < 4; ++r) h1[c][r] = fmaxf(s.acc1[c][r] + h0[c][r], 0.0f);
            if (writer) {
                if constexpr (HB == 0) { for (int r = 0; r < 4; ++r) H1[hw + (r << 6) + c] = h1[c][r]; }
                else store_quad16<HB>(H1, T, c, p, h1[c]);
            }
        });
        if constexpr (last) publish(flags, F_H1, gen + T + 1);
        if constexpr (I == 0) {
            float4 v[4];
            take(flags, F_P2, gen + T + 1, [&]() __attribute__((always_inline)) {
                for (int a = 0; a < 4; ++a) v[a] = *reinterpret_cast<const float4*>(SP2 + sub_off_sp + 4 * a);
            });
            for (int a = 0; a < 4; ++a) s.acc2[a] = f32x4{v[a].x, v[a].y, v[a].z, v[a].w};
        }
        BLOCKS(acc2, w2, h1, c0, c1 + 1)
        if constexpr (nx < 4) { BLOCKS(acc2, w2, h1, nx, nx + 1) }
        // ---- layer 2
        sfor<c0, c1 + 1>([&](auto cc) __attribute__((always_inline)) {
            constexpr int c = decltype(cc)::value;
            for (int r = 0; r < 4; ++r) h2[c][r] = fmaxf(s.acc2[c][r] + h1[c][r], 0.0f);
            if (writer) {
                if constexpr (HB == 0) { for (int r = 0; r < 4; ++r) H2[hw + (r << 6) + c] = h2[c][r]; }
                else store_quad16<HB>(H2, T, c, p, h2[c]);
            }
        });
        if constexpr (last) publish(flags, F_H2, gen + T + 1);
        // ---- output rows of this tile's ranks: (shift, raw) of groups (0,1) in o[0], (2,3) in o[1]
        if constexpr (I == 0) {
            const int O0 = (s.g[0] < D ? s.g[0] : 0) >> 3;
            float2 po[4];
            take(flags, F_P3, gen + T + 1, [&]() __attribute__((always_inline)) {
                for (int i = 0; i < 4; ++i) {
                    const int gg = s.g[i] < D ? s.g[i] : 0;
                    const int slot = ((gg >> 3) != O0) ? 1 : 0;
                    po[i] = *reinterpret_cast<const float2*>(SP3 + slot * (sp3_rows * SPAD) + sub_off_sp + 2 * (gg & 7));
                }
            });
            s.o[0] = f32x4{po[0].x, po[0].y, po[1].x, po[1].y};
            s.o[1] = f32x4{po[2].x, po[2].y, po[3].x, po[3].y};
        }
        constexpr int slot = I >> 1;
        sfor<c0, c1 + 1>([&](auto bb) __attribute__((always_inline)) {
            constexpr int b = decltype(bb)::value;
            sfor<0, 4>([&](auto kk) __attribute__((always_inline)) {
                constexpr int k = decltype(kk)::value;
                s.o[slot] = M4<4 * b + k>(slot ? s.w3.y : s.w3.x, h2[b][k], s.o[slot]);
            });
        });
        if constexpr (slot == 0 && NG > 2) {              // gap: the same quads into the rows of groups 2, 3
            sfor<c0, c1 + 1>([&](auto bb) __attribute__((always_inline)) {
                constexpr int b = decltype(bb)::value;
                sfor<0, 4>([&](auto kk) __attribute__((always_inline)) {
                    constexpr int k = decltype(kk)::value;
                    s.o[1] = M4<4 * b + k>(s.w3.y, h2[b][k], s.o[1]);
                });
            });
        }
        const float shift = s.o[slot][2 * (I & 1)];
        const float ls = fast_ls(s.o[slot][2 * (I & 1) + 1]);
        float xg = (s.yv[I] - shift) * fast_exp_neg(ls);
        xg = live ? xg : 0.0f;
        ladj -= live ? ls : 0.0f;
        if (writer && live) store_x<HB>(X, X16, g, p, xg);
        if constexpr (last) publish(flags, F_X, gen + T + 1);
        // ---- layer-0 column of the new rank: the next quad first (k slot 4 + I of this tile's window, or, after the
        // tile's last group, k slot I of the next tile's), then everything that feeds later quads
        if constexpr (nx < 4) s.a0[nx] = M4<4 + I>(comp(s.w0, nx), xg, s.a0[nx]);
        s.a0n[0] = M4<I>(comp(s.w0n, 0), xg, s.a0n[0]);
        if constexpr (nx + 1 < 4) { BLOCKS(acc1, w1, h0, nx + 1, 4) BLOCKS(acc2, w2, h1, nx + 1, 4) }
        sfor<nx + 1, 4>([&](auto aa) __attribute__((always_inline)) {
            constexpr int a = decltype(aa)::value;
            s.a0[a] = M4<4 + I>(comp(s.w0, a), xg, s.a0[a]);
        });
        sfor<1, 4>([&](auto aa) __attribute__((always_inline)) {
            constexpr int a = decltype(aa)::value;
            s.a0n[a] = M4<I>(comp(s.w0n, a), xg, s.a0n[a]);
        });
#undef BLOCKS
        chain_group6<PAT, I + 1, HB>(s, H0, H1, H2, X, X16, flags, SP1, SP2, SP3, T, D, gen, sub_off_sp, sp3_rows, p, writer, ladj);
    }
}

}  // namespace tri6

// NS: 16-walker subsets per workgroup (1, 2 or 4); FM: 0 = plain inverse of `in`, 4 / 8 / 16 / 32 = fused proposal, D <= 4 FM;
// HB: operand type of the helpers' left-looking products (0 float32, 1 bfloat16, 2 float16; Ops above).
// Registers (round 5).  Left to itself the compiler gives a four-wavefront instance 300-360 registers: the accumulators of the
// chain's MFMAs go to AGPRs and every value the vector ALU needs of them comes back through v_accvgpr_read (1068 such moves
// in the one-subset 16-bit instance, 2852 with two subsets, none in the five-wavefront instances, which have to fit 256) --
// on a wavefront that is bound by instruction issue.  The plain-inverse instances whose state fits are therefore held to
// 256 architectural registers (second launch-bound argument: two wavefronts per SIMD): no AGPR, no scratch, and the
// 16-bit sweep of 5000 x 128 walkers goes from 755 to 649 us (one subset 715 -> 645; D = 50 / maf6 381 -> 329).  With two
// subsets the helpers' chunks are two pairs instead of four (below: CH) -- they wait thousands of cycles per tile for
// the chain anyway.  Four subsets and the fused instances need more than 256 and keep the default.
template <int NS, int FM, int NW = 4, int HB = 0>
__global__ __launch_bounds__(64 * NW, (FM == 0 && (NS == 1 || (NS == 2 && HB != 0))) ? 2 : 1) void maf_inverse_tri6_kernel(pmc_maf_t m, const float* __restrict__ in,
                                                               float* __restrict__ out, float* __restrict__ ladj_out,
                                                               int64_t n, ProposeArgs pa) {
    using namespace tri6;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, p = lane & 15;
    const int D = m.D, Dp = m.Dp, Hp = m.Hp, Tn = m.T, nT = m.nT, nXT = m.nXT, nOT = m.nOT;
    // ---- LDS map
    using HO = Ops<HB>;
    constexpr int SP3R = 16 * NS;                         // walker rows of one staged output tile
    // Yb: the transform's input by rank; Xb: its solution by rank (zeros where not solved yet: the layer-0 helper multiplies
    // whole tiles).  With 16-bit helpers the helper reads X16b instead (x once more, as B operand of the 16x16x16 MFMA) and
    // Xb IS Yb: x_g overwrites y_g in place (a rank's y is read before its x is written) -- the 8 KB per subset that let a
    // second subset share the workgroup at D = 128.
    const int szY = Dp * 16, szX16 = HB ? HO::act_floats(nXT) : 0, szH = HO::act_floats(nT);      // floats per subset
    float* Yb = smem;                                    // [NS][szY]
    float* Xb = HB ? Yb : Yb + NS * szY;                 // [NS][szY]
    float* X16b = Xb + NS * szY;                         // [NS][szX16]
    float* H0b = X16b + NS * szX16;                      // [NS][szH]
    float* H1b = H0b + NS * szH;
    float* H2b = H1b + NS * szH;
    float* SP0 = H2b + NS * szH;                         // [2 parities][NS][16][SPAD]
    float* SP1 = SP0 + 2 * NS * 16 * SPAD;
    float* SP2 = SP1 + 2 * NS * 16 * SPAD;
    float* SP3 = SP2 + 2 * NS * 16 * SPAD;               // [2 parities][2 output tiles][16 NS][SPAD]
    int* flags = reinterpret_cast<int*>(SP3 + 2 * 2 * SP3R * SPAD);
    const int* feat_of_rank = m.meta + 8;
    const int* rank_of_feat = m.meta + 8 + Tn * D;
    const int* quad_meta = m.meta + 8 + 2 * Tn * D;

    const int oF0 = 0;
    const int oF1 = oF0 + nT * nXT * 1024;
    const int oF2 = oF1 + nT * nT * 1024;
    const int oF3 = oF2 + nT * nT * 1024;
    const int oW0 = oF3 + nOT * nT * 1024;
    const int oB0 = oW0 + Dp * Hp * 4;
    const int oB1 = oB0 + Hp * 4;
    const int oB2 = oB1 + Hp * 4;
    const int oB3 = oB2 + Hp * 4;
    const int oCW1 = oB3 + nOT * 64;
    const int oCW2 = oCW1 + nT * 1024;
    const int oCW0 = oCW2 + nT * 1024;
    const int oCW3 = oCW0 + nT * 1024;
    const int oF0C = oCW3 + nT * 512;
    const int blk_bytes = (int)(m.pk_per_transform * 4);
    const int vo_lane = lane << 4, vo_q = q << 4;
    (void)oF0; (void)oW0;
    // sections of the helpers' fragment image: the float32 image itself, or the 16-bit one (f1 | f2 | f3, 512-byte blocks)
    const int rbH = HO::row_bytes(nT), rbX = HO::row_bytes(nXT);      // bytes of a fragment row over the hidden / the rank tiles
    const int hF1 = HB ? 0 : oF1, hF2 = HB ? nT * rbH : oF2, hF3 = HB ? 2 * nT * rbH : oF3;
    const int hF0C = HB ? (2 * nT + nOT) * rbH : oF0C;
    const int h16_bytes = (2 * nT + nOT) * rbH + nT * rbX;

    const int64_t set0 = (int64_t)blockIdx.x * NS;       // first 16-walker set of this workgroup
    if (threadIdx.x < F_COUNT) flags[threadIdx.x] = 0;
    if constexpr (NW == 5) {
        // Five wavefronts on four SIMDs: two of them share one.  Roles go by what a shared SIMD costs: the chain (role 0)
        // and the output rows (3) to the wavefronts that have a SIMD to themselves, then the hidden layers (1, 2), the
        // layer-0 partials (4: the lightest) last.
        int* simd_of = flags + F_COUNT;
        if (lane == 0) simd_of[wv] = (int)((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 4) & 3);   // HW_ID.SIMD_ID
        __syncthreads();
        int sd[5], sharers[5];
        for (int w = 0; w < 5; ++w) sd[w] = __builtin_amdgcn_readfirstlane(simd_of[w]);
        for (int w = 0; w < 5; ++w) { sharers[w] = 0; for (int v = 0; v < 5; ++v) sharers[w] += sd[v] == sd[w]; }
        int before = 0;                                  // wavefronts that choose before this one: fewer sharers, then lower index
        for (int v = 0; v < 5; ++v) before += sharers[v] < sharers[wv] || (sharers[v] == sharers[wv] && v < wv);
        const int order[5] = {0, 3, 1, 2, 4};
        int role = 4;
        for (int i = 0; i < 5; ++i) if (before == i) role = order[i];
        wv = role;
    }
    // ---- input: proposals (fused) or rows of `in`, one subset per wavefront
    for (int sb = wv; sb < NS; sb += 4) {
        float* Y = Yb + sb * szY;
        if constexpr (FM > 0) {
            for (int e = lane; e < (Dp - D) * 16; e += 64) Y[lidx(D + (e >> 4), e & 15)] = 0.0f;
            const double sg = pa.adapt ? pa.adapt[0] : pa.sigma, ca = pa.adapt ? pa.adapt[1] : pa.cn_a;
            propose_body<FM>(pa.kind, pa.cur32, nullptr, pa.adapt ? pa.adapt + 2 : pa.mu, pa.inv_cov, pa.chol, pa.nu, sg,
                             ca, pa.rng, pa.prop64, nullptr, pa.quad, pa.quad_prop, n, D, Y, rank_of_feat + (Tn - 1) * D,
                             set0 + sb);
        } else {
            load_rows(Y, in, (set0 + sb) * 16, n, D, Dp, feat_of_rank + (Tn - 1) * D, lane);
        }
    }
    float ladj = 0.0f;
    // chain lanes: subset (lane >> 4) % NS, walker p; lanes beyond 16 NS shadow the first ones and do not write
    const int csub = (lane >> 4) % NS;
    const bool writer = lane < 16 * NS;
    const int sub_sp = (csub * 16 + p) * SPAD;            // this lane's row inside a staging tile
    // the tiles' rank words, once: lane T holds tile T's (v_readlane per tile; nT <= 64)
    int4 dgl = make_int4(D, D, D, D);
    if (lane < nT) dgl = *reinterpret_cast<const int4*>(quad_meta + 4 * lane);
    dgl.x &= 0xffff; dgl.y &= 0xffff; dgl.z &= 0xffff; dgl.w &= 0xffff;

    int gen = 1;                                         // (words start at 0: nothing is published)
    for (int t = Tn - 1; t >= 0; --t, gen += 256) {
        const float* blk = m.packed + (size_t)t * m.pk_per_transform;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)blk, 0, blk_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsw = HB ? __builtin_amdgcn_make_buffer_rsrc(
            (void*)(reinterpret_cast<const unsigned char*>(m.lane16) + (size_t)t * h16_bytes), 0, h16_bytes, 0x00020000) : rs;
        {   // ranks not solved yet are zeros where the layer-0 helper reads x (it multiplies whole tiles)
            float4* z4 = reinterpret_cast<float4*>(HB ? X16b : Xb);
            for (int e = threadIdx.x; e < (NS * (HB ? szX16 : szY)) >> 2; e += 64 * NW) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();

        if (wv == 0) {
            // ================================================================== CHAIN
            ChainState s;
            float* H0 = H0b + csub * szH; float* H1 = H1b + csub * szH; float* H2 = H2b + csub * szH;
            float* X = Xb + csub * szY;
            float* X16 = X16b + csub * szX16;
            const float* Y = Yb + csub * szY;
            auto tile_words = [&](int T) __attribute__((always_inline)) {
                int4 dg;
                dg.x = __builtin_amdgcn_readlane(dgl.x, T); dg.y = __builtin_amdgcn_readlane(dgl.y, T);
                dg.z = __builtin_amdgcn_readlane(dgl.z, T); dg.w = __builtin_amdgcn_readlane(dgl.w, T);
                return dg;
            };
            // rank 0 reads nothing: bias only
            float x0;
            {
                const float* b3 = blk + (oB3 >> 2);
                const float shift = b3[0], ls = fast_ls(b3[1]);
                x0 = (Y[lidx(0, p)] - shift) * fast_exp_neg(ls);
                ladj -= ls;
                if (writer) store_x<HB>(X, X16, 0, p, x0);
            }
            publish(flags, F_X, gen + 0);
            float4 w1n = bload4(rs, vo_lane, oCW1), w2n = bload4(rs, vo_lane, oCW2), w0nn = bload4(rs, vo_lane, oCW0);
            float2 w3n = bload2(rs, lane << 3, oCW3);
            for (int a = 0; a < 4; ++a) s.a0n[a] = f32x4{0.f, 0.f, 0.f, 0.f};
            // rank 0 is "group 0 of the tile before tile 0": k slot 0 of tile 0's window
            sfor<0, 4>([&](auto aa) __attribute__((always_inline)) {
                constexpr int a = decltype(aa)::value;
                s.a0n[a] = M4<0>(comp(w0nn, a), x0, s.a0n[a]);
            });
            for (int T = 0; T < nT; ++T) {
                // (the tile's rank words are the same in every lane and come through v_readlane: what follows from them stays in
                // scalar registers -- otherwise every buffer load whose offset depends on a rank is wrapped in a waterfall loop)
                const int4 dg = tile_words(T);
                if (dg.x >= D && dg.y >= D && dg.z >= D && dg.w >= D) break;       // padding tiles
                const int pat = 1 | ((dg.y != dg.x) << 1) | ((dg.z != dg.y) << 2) | ((dg.w != dg.z) << 3);
                tile_groups(dg, D, s.g);
                long long* pf = (pa.prof && blockIdx.x == 0) ? pa.prof + ((size_t)((Tn - 1 - t) * nT + T) * 4 + 0) * 4 : nullptr;
                if (pf && lane == 0) { pf[0] = clock64(); if (T == 0) pf[3] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4); }
                s.w1 = w1n; s.w2 = w2n; s.w0 = w0nn; s.w3 = w3n;
                if (T + 1 < nT) {
                    w1n = bload4(rs, vo_lane, oCW1 + (T + 1) * 1024);
                    w2n = bload4(rs, vo_lane, oCW2 + (T + 1) * 1024);
                    w0nn = bload4(rs, vo_lane, oCW0 + (T + 1) * 1024);
                    w3n = bload2(rs, lane << 3, oCW3 + (T + 1) * 512);
                } else {
                    w0nn = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                s.w0n = w0nn;
                for (int i = 0; i < 4; ++i) s.yv[i] = Y[lidx(s.g[i] < D ? s.g[i] : 0, p)];
                // layer-0 partials of everything the helper multiplied (ranks before the previous tile's) + what the
                // chain accumulated for this tile while it ran the previous one
                {
                    const float* sp = SP0 + (T & 1) * (NS * 16 * SPAD) + sub_sp;
                    float4 v[4];
                    take(flags, F_P0, gen + T + 1, [&]() __attribute__((always_inline)) {
                        for (int a = 0; a < 4; ++a) v[a] = *reinterpret_cast<const float4*>(sp + 4 * a);
                    });
                    for (int a = 0; a < 4; ++a) {
                        s.a0[a] = f32x4{v[a].x + s.a0n[a][0], v[a].y + s.a0n[a][1], v[a].z + s.a0n[a][2], v[a].w + s.a0n[a][3]};
                        s.a0n[a] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                if (pf && lane == 0) pf[1] = clock64();
                const float* sp1 = SP1 + (T & 1) * (NS * 16 * SPAD);
                const float* sp2 = SP2 + (T & 1) * (NS * 16 * SPAD);
                const float* sp3 = SP3 + (T & 1) * (2 * SP3R * SPAD);
                switch (pat) {
#define CASE(P) case P: chain_group6<P, 0, HB>(s, H0, H1, H2, X, X16, flags, sp1, sp2, sp3, T, D, gen, sub_sp, SP3R, p, writer, ladj); break;
                    CASE(1) CASE(3) CASE(5) CASE(7) CASE(9) CASE(11) CASE(13) CASE(15)
#undef CASE
                }
                if (pf && lane == 0) pf[2] = clock64();
            }
        } else {
            // ================================================================== HELPERS
            // wv 1: hidden layer 1 (reads H0), wv 2: hidden layer 2 (reads H1), wv 3: layer 0 (cut) + output rows (reads X, H2)
            const float* Hin = wv == 1 ? H0b : (wv == 2 ? H1b : H2b);
            const int oF = wv == 1 ? hF1 : hF2, oB = wv == 1 ? oB1 : oB2;
            float* SP = wv == 1 ? SP1 : SP2;
            const int f_in = wv == 1 ? F_H0 : (wv == 2 ? F_H1 : F_H2);
            const int f_out = wv == 1 ? F_P1 : F_P2;
            int known = gen;                              // tiles < known - gen of the input layer are final
            auto need = [&](int K) {                       // tile K of the input layer must be final
                if (known < gen + K + 1) { wait_for<HB != 0>(flags, f_in, gen + K + 1); known = peek(flags, f_in); }
            };
            // tiles per chunk of the pipelined left-looking products (two chunks' fragments in flight): 16-bit fragments are a
            // quarter of the matrix-pipe time per tile, so the L2 latency needs twice the tiles in flight to stay hidden
            // (two subsets with 16-bit operands: two pairs per chunk keep the instance within 256 registers, see the kernel's header)
            constexpr int CH = (NS == 1 || (NS == 2 && HB == 0)) ? 4 : 2;
            // wave 3 keeps the sums of two output tiles (slot = tile & 1) across the hidden tiles: consecutive hidden tiles
            // share their output tiles (8 ranks each), so a hidden tile adds only the h2 tiles that became final since the
            // slot was last staged instead of summing from tile 0 again (same order of additions, same bits); a new output
            // tile costs one full row, once.
            f32x4 oacc0[NS], oacd0[NS], oacc1[NS], oacd1[NS];
            int slotO0 = -1, slotO1 = -1, slotK0 = 0, slotK1 = 0;
            float4 obias0 = make_float4(0.f, 0.f, 0.f, 0.f), obias1 = obias0;
            // the wavefront that forms the layer-0 partials: the fifth; of four, the output wavefront -- or, with 16-bit operands,
            // the hidden layer 2's: its row is staged when the chain starts the tile, and it idles from there until the tile's h1
            // is final, while the output wavefront (a tile's output partials, then a new output tile's whole row) is the one the
            // chain waits for (measured: 6.5-8.5 k cycles per iteration against the chain's 5.5 k)
            constexpr int P0W = NW == 5 ? 4 : ((HB && !(TRI6_ABL & 4)) ? 2 : 3);
            if (wv == P0W) {
                // layer-0 partial of tile 0: bias only (cut = 0)
                const float4 b0 = bload4(rs, vo_q, oB0);
                for (int sb = 0; sb < NS; ++sb)
                    *reinterpret_cast<float4*>(SP0 + (sb * 16 + p) * SPAD + 4 * q) = b0;
                publish(flags, F_P0, gen + 1);
            }
            // the tiles' rank words: lane T loads tile T's once per transform, a tile reads them with v_readlane (a global
            // load + readfirstlane per tile put an L2 round trip at the head of every helper tile); live tiles are a prefix
            const int nTr = __builtin_amdgcn_readfirstlane(
                __builtin_popcountll(__builtin_amdgcn_ballot_w64(dgl.x < D || dgl.y < D || dgl.z < D || dgl.w < D)));
            // ---- layer-0 partial of the NEXT tile: ranks before this tile's own (f0c: the fragments right of the cut are zeros), final
            // once tile T-1 is (T = 0: rank 0 is there; the wait sits behind the first fragments' loads).  Pasted where the wavefront that
            // forms it (P0W) has its tile loop -- as a lambda it moved the five-wave variant past its 256 registers.
#define LAYER0_STEP() \
                    if (T + 1 < nT) { \
                        const float4 b0 = bload4(rs, vo_q, oB0 + 64 * (T + 1)); \
                        f32x4 acc[NS], acd[NS]; \
                        for (int sb = 0; sb < NS; ++sb) { acc[sb] = f32x4{0.f, 0.f, 0.f, 0.f}; acd[sb] = f32x4{0.f, 0.f, 0.f, 0.f}; } \
                        const int nXn = (gfirst + 15) >> 4; \
                        if (nXn > 0) \
                            left_products<HB, NS, CH>(acc, acd, rsw, hF0C + (T + 1) * rbX, HB ? X16b : Xb, HB ? szX16 : szY, 0, nXn < nXT ? nXn : nXT, lane, \
                                                      [&]() __attribute__((always_inline)) { if (pe && lane == 0) pe[4] = clock64(); wait_for<HB != 0>(flags, F_X, gen + T); if (pe && lane == 0) pe[5] = clock64(); }); \
                        if (pe && lane == 0) pe[1] = clock64(); \
                        float* sp0 = SP0 + ((T + 1) & 1) * (NS * 16 * SPAD); \
_Pragma("unroll") \
                        for (int sb = 0; sb < NS; ++sb) \
                            *reinterpret_cast<float4*>(sp0 + (sb * 16 + p) * SPAD + 4 * q) = \
                                make_float4((acc[sb][0] + acd[sb][0]) + b0.x, (acc[sb][1] + acd[sb][1]) + b0.y, \
                                            (acc[sb][2] + acd[sb][2]) + b0.z, (acc[sb][3] + acd[sb][3]) + b0.w); \
                        publish(flags, F_P0, gen + T + 2); \
                        if (pf && lane == 0 && wv == 3) pf[3] = clock64(); \
                    }
            if (wv == 1 || wv == 2) {
            // hidden-layer helpers: what the current row needs first -- requested a tile ahead (row 0: its bias only)
            using WC = std::conditional_t<HB == 0, float4, uint4>;         // a chunk element: a tile's block / a pair of tiles
            WC hw0[CH], hw1[CH];
            typename HO::W hwl = HO::zero();
            float4 hbb = make_float4(0.f, 0.f, 0.f, 0.f);
            if (wv == 1 || wv == 2) hbb = bload4(rs, vo_q, oB);
            for (int T = 0; T < nTr; ++T) {
                // (uniform, and said so: what follows from the rank words stays in scalar registers -- otherwise every buffer
                // load whose offset depends on a rank is wrapped in a waterfall loop)
                int4 dg;
                dg.x = __builtin_amdgcn_readlane(dgl.x, T); dg.y = __builtin_amdgcn_readlane(dgl.y, T);
                dg.z = __builtin_amdgcn_readlane(dgl.z, T); dg.w = __builtin_amdgcn_readlane(dgl.w, T);
                long long* pf = (pa.prof && blockIdx.x == 0 && wv < 4) ? pa.prof + ((size_t)((Tn - 1 - t) * nT + T) * 4 + wv) * 4 : nullptr;
                if (pf && lane == 0) { pf[0] = clock64(); if (T == 0) pf[2] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4); }
                const int gfirst = dg.x;
                if (wv == 1 || wv == 2) {
                    f32x4 acc[NS], acd[NS];
                    // (the bias joins the sum when it is staged: as the accumulator's first value its load would have to land
                    // before the first fragment is even requested)
                    for (int sb = 0; sb < NS; ++sb) { acc[sb] = f32x4{0.f, 0.f, 0.f, 0.f}; acd[sb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                    const int base_ = oF + T * rbH;
                    // acc/acd[sb] += sum_{K < T} frag[K] . Hin[sb][K]: the tiles before the last are final as soon as the one
                    // before the last is (pipelined: left_products), the last tile of the input layer is awaited right before
                    // its use; this row's first fragments, its last one and its bias were requested a tile ago (hw0, hw1, hwl,
                    // hbb), the NEXT row's are requested before the wait
                    if (T > 1) {
                        if constexpr (HB == 0)
                            left32_w<NS, CH, true>(acc, acd, rsw, base_, Hin, szH, T - 1, lane, hw0, hw1,
                                                   [&]() __attribute__((always_inline)) { need(T - 2); });
                        else
                            left16_w<HB, NS, CH, true>(acc, acd, rsw, base_, Hin, szH, 0, T - 1, lane, hw0, hw1,
                                                       [&]() __attribute__((always_inline)) { need(T - 2); });
                    }
                    float4 nbb = hbb;
                    typename HO::W nwl = hwl;
                    if (T + 1 < nTr) {
                        const int basen = base_ + rbH;
                        nbb = bload4(rs, vo_q, oB + 64 * (T + 1));
                        nwl = HO::loadw_tile(rsw, lane, basen, T);
                        if (T > 0) {
                            if constexpr (HB == 0) first_chunks32<CH>(hw0, hw1, rsw, lane, basen, T);
                            else first_chunks16<CH>(hw0, hw1, rsw, lane, basen, T);
                        }
                        asm volatile("" ::: "memory");
                    }
                    if (T > 0) {
                        if (pf && lane == 0) pf[1] = clock64();
                        need(T - 1);
                        if (pf && lane == 0) pf[2] = clock64();
#pragma unroll
                        for (int sb = 0; sb < NS; ++sb)
                            HO::mma(acc[sb], acd[sb], hwl, HO::loadb(Hin + sb * szH, T - 1, lane), 1);
                    }
                    float* sp = SP + (T & 1) * (NS * 16 * SPAD);
#pragma unroll
                    for (int sb = 0; sb < NS; ++sb)
                        *reinterpret_cast<float4*>(sp + (sb * 16 + p) * SPAD + 4 * q) =
                            make_float4((acc[sb][0] + acd[sb][0]) + hbb.x, (acc[sb][1] + acd[sb][1]) + hbb.y,
                                        (acc[sb][2] + acd[sb][2]) + hbb.z, (acc[sb][3] + acd[sb][3]) + hbb.w);
                    publish(flags, f_out, gen + T + 1);
                    if (pf && lane == 0) pf[3] = clock64();
                    hbb = nbb; hwl = nwl;
                    if constexpr (P0W == 2) { if (wv == 2) { long long* pe = nullptr; LAYER0_STEP() } }
                }
            }
            } else {
            for (int T = 0; T < nTr; ++T) {
                // (uniform, and said so: what follows from the rank words stays in scalar registers -- otherwise every buffer
                // load whose offset depends on a rank is wrapped in a waterfall loop)
                int4 dg;
                dg.x = __builtin_amdgcn_readlane(dgl.x, T); dg.y = __builtin_amdgcn_readlane(dgl.y, T);
                dg.z = __builtin_amdgcn_readlane(dgl.z, T); dg.w = __builtin_amdgcn_readlane(dgl.w, T);
                long long* pf = (pa.prof && blockIdx.x == 0 && wv < 4) ? pa.prof + ((size_t)((Tn - 1 - t) * nT + T) * 4 + wv) * 4 : nullptr;
                if (pf && lane == 0) { pf[0] = clock64(); if (T == 0) pf[2] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4); }
                long long* pe = (pf && wv == 3) ? pa.prof + (size_t)Tn * nT * 16 + (size_t)((Tn - 1 - t) * nT + T) * 8 : nullptr;
                int gfirst = dg.x, glast = dg.x;
                if (dg.y < D) glast = dg.y;
                if (dg.z < D) glast = dg.z;
                if (dg.w < D) glast = dg.w;
                if (wv == 3) {
                    // ---- output partials of this tile's ranks: output tile(s) O = rank >> 3 against h2 of tiles < T
                    const int O0 = gfirst >> 3, O1 = glast >> 3;
                    float* sp3 = SP3 + (T & 1) * (2 * SP3R * SPAD);
                    // first everything that does not wait for the chain's current tile (a new output tile's row up to
                    // tile T - 2), then the last tile of both
                    const int nso = O1 != O0 ? 2 : 1;
                    typename HO::W wl[2];
                    auto pre = [&](f32x4 (&acc)[NS], f32x4 (&acd)[NS], int& sO, int& sK, float4& ob, int O, typename HO::W& w) __attribute__((always_inline)) {
                        sO = __builtin_amdgcn_readfirstlane(sO); sK = __builtin_amdgcn_readfirstlane(sK);
                        if (sO != O) {
                            ob = bload4(rs, vo_q, oB3 + 64 * O);
                            for (int sb = 0; sb < NS; ++sb) { acc[sb] = f32x4{0.f, 0.f, 0.f, 0.f}; acd[sb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                            sO = O; sK = 0;
                        }
                        const int base_ = hF3 + O * rbH;
                        if (T > sK) w = HO::loadw_tile(rsw, lane, base_, T - 1);
                        if (T - 1 > sK)
                            left_products<HB, NS, CH>(acc, acd, rsw, base_, Hin, szH, sK, T - 1, lane,
                                                      [&]() __attribute__((always_inline)) { need(T - 2); });
                    };
                    auto fin = [&](f32x4 (&acc)[NS], f32x4 (&acd)[NS], int& sK, const float4& ob, int so, const typename HO::W& w) __attribute__((always_inline)) {
                        if (T > sK) {
#pragma unroll
                            for (int sb = 0; sb < NS; ++sb)
                                HO::mma(acc[sb], acd[sb], w, HO::loadb(Hin + sb * szH, T - 1, lane), 1);
                            sK = T;
                        }
#pragma unroll
                        for (int sb = 0; sb < NS; ++sb)
                            *reinterpret_cast<float4*>(sp3 + so * (SP3R * SPAD) + (sb * 16 + p) * SPAD + 4 * q) =
                                make_float4((acc[sb][0] + acd[sb][0]) + ob.x, (acc[sb][1] + acd[sb][1]) + ob.y,
                                            (acc[sb][2] + acd[sb][2]) + ob.z, (acc[sb][3] + acd[sb][3]) + ob.w);
                    };
                    for (int so = 0; so < nso; ++so) {
                        const int O = O0 + so;
                        if (O & 1) pre(oacc1, oacd1, slotO1, slotK1, obias1, O, wl[1]); else pre(oacc0, oacd0, slotO0, slotK0, obias0, O, wl[0]);
                    }
                    if (pf && lane == 0 && T > 0) pf[2] = clock64();
                    if (T > 0) need(T - 1);
                    if (pe && lane == 0) pe[0] = clock64();
                    for (int so = 0; so < nso; ++so) {
                        const int O = O0 + so;
                        if (O & 1) fin(oacc1, oacd1, slotK1, obias1, so, wl[1]); else fin(oacc0, oacd0, slotK0, obias0, so, wl[0]);
                    }
                    publish(flags, F_P3, gen + T + 1);
                    if (pf && lane == 0) pf[1] = clock64();
                }
                if (wv == P0W && P0W != 2) { LAYER0_STEP() }
                if (wv == 3) {
                    if (T + 1 < nT) {
                        // ---- ahead of the chain: a NEW output tile of the next hidden tile starts its row now, over the
                        // h2 tiles that are final (all before this one), while the chain works through this tile -- the
                        // next tile then only adds one h2 tile to each of its sums
                        int4 dn;
                        dn.x = __builtin_amdgcn_readlane(dgl.x, T + 1); dn.y = __builtin_amdgcn_readlane(dgl.y, T + 1);
                        dn.z = __builtin_amdgcn_readlane(dgl.z, T + 1); dn.w = __builtin_amdgcn_readlane(dgl.w, T + 1);
                        if (T + 1 < nTr && dn.x < D) {
                            int nlast = dn.x;
                            if (dn.y < D) nlast = dn.y;
                            if (dn.z < D) nlast = dn.z;
                            if (dn.w < D) nlast = dn.w;
                            auto ahead = [&](f32x4 (&acc)[NS], f32x4 (&acd)[NS], int& sO, int& sK, float4& ob, int O) __attribute__((always_inline)) {
                                sO = __builtin_amdgcn_readfirstlane(sO); sK = __builtin_amdgcn_readfirstlane(sK);
                                if (sO == O) return;
                                ob = bload4(rs, vo_q, oB3 + 64 * O);
                                for (int sb = 0; sb < NS; ++sb) { acc[sb] = f32x4{0.f, 0.f, 0.f, 0.f}; acd[sb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                                sO = O;
                                left_products<HB, NS, CH>(acc, acd, rsw, hF3 + O * rbH, Hin, szH, 0, T, lane, []() {});
                                sK = T;
                            };
                            for (int O = dn.x >> 3; O <= (nlast >> 3); ++O) {
                                if (O & 1) ahead(oacc1, oacd1, slotO1, slotK1, obias1, O); else ahead(oacc0, oacd0, slotO0, slotK0, obias0, O);
                            }
                            if (pe && lane == 0) pe[3] = clock64();
                        }
                    }
                }
            }
            }
        }
#undef LAYER0_STEP
        __syncthreads();                                  // (the helpers are done with this transform's x and activations)
        // re-rank for the next transform (or write out) with every thread of the workgroup: the target rank of a rank is two
        // dependent global loads, so eight elements' worth are requested together (one wavefront walking the subset's
        // 2048 elements one load pair at a time took 20 k cycles per transform at D = 128).  16-bit helpers: x sits where y
        // did, so it first moves to the (now idle) activation arrays and is re-ranked from there.
        {
            const bool lastT = (t == 0);
            const int* for_cur = feat_of_rank + t * D;
            const int* rank_next = lastT ? nullptr : rank_of_feat + (t - 1) * D;
            const int per = Dp * 16, total = NS * per;
            const float* Xs = Xb;
            if constexpr (HB != 0) {
                if (rank_next) {
                    for (int e = threadIdx.x; e < (NS * szY) >> 2; e += 64 * NW)
                        reinterpret_cast<float4*>(H0b)[e] = reinterpret_cast<const float4*>(Xb)[e];
                    Xs = H0b;
                    __syncthreads();
                }
            }
            for (int e0 = threadIdx.x; e0 < total; e0 += 8 * 64 * NW) {
                int dst[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = e0 + j * 64 * NW;
                    const int r = (e % per) >> 4;
                    dst[j] = -1;
                    if (e < total && r < D) dst[j] = for_cur[r];
                }
                if (rank_next) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (dst[j] >= 0) dst[j] = rank_next[dst[j]];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = e0 + j * 64 * NW;
                    if (e >= total) continue;
                    const int sb = e / per, r = (e % per) >> 4, pp = e & 15;
                    if (r < D) {
                        const float v = Xs[sb * szY + lidx(r, pp)];
                        const int64_t row = (set0 + sb) * 16 + pp;
                        if (rank_next) Yb[sb * szY + lidx(dst[j], pp)] = v;
                        else if (row < n) out[row * D + dst[j]] = v;
                    } else if (rank_next) {
                        Yb[sb * szY + lidx(r, pp)] = 0.0f;        // padding ranks
                    }
                }
            }
        }
        __syncthreads();
    }
    if (wv == 0 && ladj_out && writer) {
        const int64_t row = set0 * 16 + lane;
        if (row < n) ladj_out[row] = ladj;
    }
}

static int tri6_hb(const pmc_maf_t* m) { return (m->lane16 && (m->lane16_fmt == 1 || m->lane16_fmt == 2)) ? m->lane16_fmt : 0; }

static size_t tri6_lds_bytes(const pmc_maf_t* m, int ns, int hb = 0) {
    const int h_floats = hb ? ((m->nT + 1) >> 1) * 256 : m->nT * 256;       // (Ops<HB>::act_floats: one activation array of a subset)
    const int x_floats = hb ? m->Dp * 16 + ((m->nXT + 1) >> 1) * 256 : 2 * m->Dp * 16;   // y, x by rank (16-bit helpers: x over y in place + the helper's copy)
    return (size_t)(ns * (x_floats + 3 * h_floats) + 3 * 2 * ns * 16 * tri6::SPAD + 2 * 2 * 16 * ns * tri6::SPAD) * sizeof(float)
           + (tri6::F_COUNT + 8) * sizeof(int);          // (+ 8: the wavefronts' SIMD ids of the five-wave variant)
}

// walker subsets per workgroup: as few as keep the launch in one round (a chain wavefront takes the same time for 16
// and for 64 walkers; the helpers' share grows with the subsets), as many as the LDS admits otherwise
static int tri6_five_min() {
    static const int v = pmc_env_int("PMC_TRI6_FIVE_MIN", 16);   // (A/B runs)
    return v;
}
// the five-wavefront variant: plain float32 inverse of a flow with >= 16 hidden tiles, one or two subsets (the kernel must
// stay within 256 registers: two wavefronts share a SIMD, and only one such workgroup fits a CU).  With 16-bit helper
// operands the helpers are an order of magnitude below the chain: four wavefronts, a SIMD each.
static bool tri6_five(const pmc_maf_t* m, bool fused, int hb = 0) {
    return !fused && !hb && m->nT >= tri6_five_min() && !(m->reserved & PMC_MAF_VARIANT_LANE_FOUR);
}
static int tri6_subsets(const pmc_maf_t* m, int64_t n, bool fused, int hb = 0) {
    static const int forced = pmc_env_int("PMC_TRI6_SUBSETS", 0);
    const bool five = tri6_five(m, fused, hb);
    int best = 0;
    for (int ns = 1; ns <= (five ? 2 : 4); ns *= 2) {
        const size_t lds = tri6_lds_bytes(m, ns, hb);
        if (lds > 160 * 1024) break;
        if (forced == ns) return ns;
        best = ns;
        // one workgroup per CU: the five-wave variant by construction, the four-wave instances by their registers (264-416)
        if (!forced && (n + 16 * ns - 1) / (16 * ns) <= 256) break;
    }
    return best;
}

// AUTO's choice between this sweep and the register-chain sweeps of maf_inverse_tri4.hip for the flows both cover
// (D <= 64): with >= 16 hidden tiles the five-wavefront variant is faster -- D = 50 / maf6 (25 tiles): 314 us per round
// of <= 4096 walkers against 645-650 us of the two-wave sweep for <= 8192; D = 64 / maf3 (17 tiles): 120 against 160 us --
// below that the two-wave sweep is (D = 32 / maf3, 9 tiles: 61-64 against 66-83 us).  The step then launches the
// proposal and the scaler on their own (the fused instances of this kernel need more than 256 registers).
bool pmc_tri6_preferred(const pmc_maf_t* m) {
    if (m->n_out != 2 || !m->tri_ok || m->pk_per_transform * 4 > 0x7fffffffLL) return false;
    const int hb = tri6_hb(m);
    if (hb) return m->nT >= tri6_five_min() && tri6_lds_bytes(m, 1, hb) <= 160 * 1024;
    return tri6_five(m, false) && tri6_lds_bytes(m, 1) <= 160 * 1024;
}

// same contract as pmc_launch_inverse_tri4 / pmc_launch_propose_inverse_tri4 (pa == nullptr: plain inverse of z);
// -1: this flow is not covered (spline flows, degree groups wider than a tile, tiles beyond the LDS).
// m->lane16 (pmc_maf_pack_lane16): the helpers multiply with 16-bit operands (Ops above).
int pmc_launch_tri6(const ProposeArgs* pa, const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n,
                    hipStream_t stream) {
    if (m->n_out != 2 || !m->tri_ok) return -1;
    if (m->pk_per_transform * 4 > 0x7fffffffLL) return -1;
    if (pa && m->D > 64) return -1;
    if (m->nT > 64) return -1;                           // (a lane per hidden tile holds its rank words)
    const int hb = tri6_hb(m);
    const int ns = tri6_subsets(m, n, pa != nullptr, hb);
    if (ns == 0) return -1;
    if (hb && m->Dp * 16 > 3 * ((m->nT + 1) >> 1) * 256) return -1;   // (the in-place re-rank parks x in the activation arrays)
    const size_t lds = tri6_lds_bytes(m, ns, hb);
    const ProposeArgs none{};
    const unsigned grid = (unsigned)((n + 16 * ns - 1) / (16 * ns));
    // wide flows (helpers saturated: their work grows with the hidden tiles, the chain's does not) get a fifth wavefront for
    // the layer-0 partials; it needs the kernel in 256 registers (two wavefronts on one SIMD): plain inverse, one subset
    const bool five = tri6_five(m, pa != nullptr, hb) && ns <= 2;
#define LAUNCH6(NSV, FMV, NWV, HBV)                                                                                \
    {                                                                                                              \
        if (lds > 48 * 1024) {                                                                                     \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(maf_inverse_tri6_kernel<NSV, FMV, NWV, HBV>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
            if (e != hipSuccess) return pmc_fail_hip(e, "hipFuncSetAttribute(maf_inverse_tri6_kernel)");           \
        }                                                                                                          \
        hipLaunchKernelGGL((maf_inverse_tri6_kernel<NSV, FMV, NWV, HBV>), dim3(grid), dim3(64 * NWV), lds, stream, *m, z, x, \
                           ladj, n, pa ? *pa : none);                                                              \
    }
#define LAUNCH6F(FMV)                                                                                              \
    { if (ns == 1) LAUNCH6(1, FMV, 4, 0) else if (ns == 2) LAUNCH6(2, FMV, 4, 0) else LAUNCH6(4, FMV, 4, 0) }
#define LAUNCH6H(HBV)                                                                                              \
    { if (ns == 1) LAUNCH6(1, 0, 4, HBV) else if (ns == 2) LAUNCH6(2, 0, 4, HBV) else LAUNCH6(4, 0, 4, HBV) }
    if (hb && pa) return -1;                             // (no fused instance with 16-bit helpers yet)
    if (hb == 1) LAUNCH6H(1)
    else if (hb == 2) LAUNCH6H(2)
    else if (five) { if (ns == 1) LAUNCH6(1, 0, 5, 0) else LAUNCH6(2, 0, 5, 0) }
    else if (!pa) LAUNCH6F(0)
    else if (m->D <= 16) LAUNCH6F(4)
    else if (m->D <= 32) LAUNCH6F(8)
    else LAUNCH6F(16)
#undef LAUNCH6H
#undef LAUNCH6F
#undef LAUNCH6
    return pmc_check_launch("maf_inverse_tri6_kernel");
}

// ---------------------------------------------------------------------------------------------------------------------
// 16-bit image of the helpers' fragments (pmc_maf_t.lane16), derived on the device from the float32 image: per transform
// the sections f1 | f2 | f3 | f0c (hidden layers 1, 2, output rows, layer 0 left of the chain's window), each [row tile][pair
// of k tiles][lane][8] (Ops above: one 16-byte load per lane feeds two tiles), a block's 256 weights W[i][k]
// re-laid from the 16x16x4 A layout (lane = (k & 3) * 16 + i, component k >> 2) to the 16x16x16 one (lane = (k >> 2) * 16
// + i, element k & 3) and rounded to nearest even.  fmt: 1 bfloat16, 2 float16.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_lane16_kernel(pmc_maf_t m, int fmt, unsigned short* __restrict__ img) {
    const int nT = m.nT, nOT = m.nOT, nXT = m.nXT;
    const int nP = (nT + 1) >> 1, nPX = (nXT + 1) >> 1;                      // pairs of a row over the hidden / the rank tiles
    const int64_t rows_h = 2 * (int64_t)nT + nOT;                           // rows of f1 | f2 | f3 (k tiles: hidden) ...
    const int64_t pairs_per_t = rows_h * nP + (int64_t)nT * nPX;            // ... | f0c (k tiles: ranks)
    // float offsets inside a transform's float32 image (MAFSpec._build_device_layout: f0 f1 f2 f3 w0n b0 b1 b2 b3 cw1 cw2 cw0 cw3 f0c)
    const int64_t off_f1 = (int64_t)nT * nXT * 256;
    const int64_t off_f0c = off_f1 + (2 * (int64_t)nT * nT + (int64_t)nOT * nT) * 256 + (int64_t)m.Dp * m.Hp + 3 * (int64_t)m.Hp +
                            (int64_t)nOT * 16 + 3 * (int64_t)nT * 256 + (int64_t)nT * 128;
    const int64_t total = pairs_per_t * m.T * 512;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t pr = e >> 9;                       // pair (512 elements: 64 lanes x 8)
        const int t = (int)(pr / pairs_per_t);
        int64_t b = pr - (int64_t)t * pairs_per_t;
        int64_t row, src0; int pair, ntiles;
        if (b < rows_h * nP) { row = b / nP; pair = (int)(b - row * nP); ntiles = nT; src0 = off_f1 + row * nT * 256; }
        else { b -= rows_h * nP; row = b / nPX; pair = (int)(b - row * nPX); ntiles = nXT; src0 = off_f0c + row * nXT * 256; }
        const int w = (int)(e & 511), lane = w >> 3, j8 = w & 7;
        const int tile = 2 * pair + (j8 >> 2), j = j8 & 3;       // lane (c, i), element j of the tile's half: k = 4 c + j
        const int c = lane >> 4, i = lane & 15;
        float v = 0.0f;
        if (tile < ntiles) v = m.packed[(size_t)t * m.pk_per_transform + (size_t)(src0 + (int64_t)tile * 256) + ((j * 16 + i) << 2) + c];
        unsigned short h;
        if (fmt == 1) { const __bf16 x = (__bf16)v; h = *reinterpret_cast<const unsigned short*>(&x); }
        else { const _Float16 x = (_Float16)v; h = *reinterpret_cast<const unsigned short*>(&x); }
        img[e] = h;
    }
}

extern "C" int64_t pmc_maf_lane16_elems(const pmc_maf_t* m) {
    if (!m) return 0;
    const int64_t nP = (m->nT + 1) >> 1, nPX = (m->nXT + 1) >> 1;
    return ((2 * (int64_t)m->nT + m->nOT) * nP + (int64_t)m->nT * nPX) * 512 * m->T;
}

extern "C" int pmc_maf_pack_lane16(const pmc_maf_t* m, int fmt, uint16_t* image, void* stream) {
    if (!m || !m->packed || !image || (fmt != 1 && fmt != 2)) return pmc_fail("pmc_maf_pack_lane16: bad argument");
    if (m->n_out != 2) return pmc_fail("pmc_maf_pack_lane16: affine flows only");
    const int64_t total = pmc_maf_lane16_elems(m);
    int64_t grid = (total + 255) / 256; if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(pack_lane16_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, *m, fmt, image);
    return pmc_check_launch("pack_lane16_kernel");
}

#if (TRI6_ABL & 32)
// measurement build only (scripts/abl_tri6.sh 32): out[2 F_COUNT] <- cycles / times the chain wavefront of workgroup 0 waited per word, then reset
extern "C" int pmc_debug_tri6_waits(unsigned long long* out) {
    unsigned long long z[2 * tri6::F_COUNT] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(tri6::g_tri6_wait), sizeof(z)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(tri6::g_tri6_wait), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
#endif

// whether PMC_INVERSE_AUTO (and with it the MCMC step) takes this sweep for the flow (bench.py names the kernel it times)
extern "C" int pmc_maf_inverse_auto_is_lane(const pmc_maf_t* m) {
    if (!m || m->n_out != 2 || !m->tri_ok) return 0;
    return (m->nOT > 8 || pmc_tri6_preferred(m)) ? 1 : 0;
}

#ifdef PMC_DEBUG_HOOKS
// measurement only (scripts/profile_tri6.py): cycle stamps of workgroup 0 -- prof[transform * nT + tile][wave 0..3][4]
extern "C" int pmc_debug_tri6_profile(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n, long long* prof,
                                      void* stream) {
    ProposeArgs pa{};
    pa.prof = prof;
    // (FM = 0 instances read nothing else of pa)
    const int hb = tri6_hb(m);
    const int ns = tri6_subsets(m, n, false, hb);
    if (ns == 0 || m->n_out != 2 || !m->tri_ok) return pmc_fail("pmc_debug_tri6_profile: flow not covered");
    const size_t lds = tri6_lds_bytes(m, ns, hb);
    const unsigned grid = (unsigned)((n + 16 * ns - 1) / (16 * ns));
    const bool five = tri6_five(m, false, hb) && ns <= 2;
#define LP(NSV, NWV, HBV)                                                                                          \
    {                                                                                                              \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(maf_inverse_tri6_kernel<NSV, 0, NWV, HBV>), \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
        hipLaunchKernelGGL((maf_inverse_tri6_kernel<NSV, 0, NWV, HBV>), dim3(grid), dim3(64 * NWV), lds, (hipStream_t)stream, *m, z, \
                           x, ladj, n, pa);                                                                        \
    }
    if (hb == 1) { if (ns == 1) LP(1, 4, 1) else if (ns == 2) LP(2, 4, 1) else LP(4, 4, 1) }
    else if (hb == 2) { if (ns == 1) LP(1, 4, 2) else if (ns == 2) LP(2, 4, 2) else LP(4, 4, 2) }
    else if (five) { if (ns == 1) LP(1, 5, 0) else LP(2, 5, 0) } else if (ns == 1) LP(1, 4, 0) else if (ns == 2) LP(2, 4, 0) else LP(4, 4, 0)
#undef LP
    return pmc_check_launch("maf_inverse_tri6_kernel<profile>");
}
#endif

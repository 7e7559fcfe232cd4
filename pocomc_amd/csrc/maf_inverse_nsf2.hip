// Triangular-sweep inverse of the neural spline flows as TWO wavefronts per 16 walkers (pocomc/mcmc.py:88 ->
// flow.py:116-132 with flow = nsf3 | nsf6 | nsf12): the spline counterpart of maf_inverse_tri5_kernel.
//
// The lone-wave spline sweep (maf_inverse_tri_nsf.hip) spends its time where the affine one did before it was split --
// 272 us for 7008 walkers of nsf3 @ D = 32, of which (timing-only builds, scripts/abl_nsf.sh) ~100 us are the rank's
// left-looking output product (23 parameters = two 16-row tiles against every final h2 tile: 8 (Tt + 1) MFMAs per
// rank), ~65 us the hidden chain with its left-looking bursts, ~53 us the spline solves, ~55 us everything else.  As
// in the affine sweep the work is cut by WHEN its inputs exist:
//   * CHAIN wave: per degree group the three hidden hops on transposed accumulators (maf_chain_rot.h: one MFMA per
//     layer and quad), its share of the NEXT tile's hidden pre-activations (right-looking: accN1 / accN2 / a0N), the
//     part of the rank's 23 parameters that comes from the previous and the own hidden tile (8 + 2 (quads so far)
//     MFMAs, fragments requested one group ahead; round 4: all of it but the group's own quads is formed in the slots
//     of the PREVIOUS rank's spline solve, see nsf_group), the exchange through a 2 KB LDS panel and the spline solve;
//   * BURST wave, a whole tile ahead: the hidden layers' left-looking products against tiles <= Tt-2 (as in tri5: f0c
//     against x, f1 / f2 against h0 / h1, into transposed staging) AND, for each of the tile's ranks, bias + the two
//     output tiles against h2 tiles <= Tt-2 (into a staged partial the chain's accumulators start from: both waves hold
//     a 16 x 16 product in the same lane layout, so staging is one 16-byte write and read per lane at the same address).
//     Round 4: the output partials of the last two live tiles start at steps 2 - 4 (eager partials, NSF2_EAGER_OK): the two
//     wavefronts are within 10 % of each other at every barrier (scripts/profile_nsf2_tiles.py).
// One LDS-only barrier per tile: E(Tt) = "tile Tt is final, the staging of tile Tt+1 is complete".
#include <stdlib.h>
#include "maf_chain_rot.h"
#include "rqs.h"
#include "propose_body.h"

#ifndef NSF2_ABL
#define NSF2_ABL 0                 // timing experiments only (scripts/abl_nsf.sh): results are wrong when != 0
#endif
#if NSF2_ABL != 0
extern "C" int pmc_ablation_nsf2(void) { return NSF2_ABL; }      // (see pmc_ablation_tri6)
#endif                             // 1 no spline solve, 2 no output MFMAs on the chain, 4 burst: no output partials, 8 nor their loads, 16 chain: no output fragment requests, 32 burst: no hidden products, 64 no eager partials (results stay right)
#define NSF2_PK 10                 // K tiles of the hidden bursts held in registers; the static burst tile covers flows of <= NSF2_PK + 1 live tiles
#define NSF2_PX 4                  // x tiles of the layer-0 product held in registers (D <= 64)
#define NSF2_OOB 0x40000000        // a lane offset beyond every image: the bounds-checked load returns zeros
#define NSF2_STAGE_FLOATS (3 * 256)                 // hidden staging S0 | S1 | S2 (transposed, [lane][4])
#define NSF2_PART_FLOATS (4 * 2 * 256)              // output staging [group][half][lane][4]
#define NSF2_TT_WORDS(m) (((m)->nT + 2) * 8)        // per-tile table: ranks (word 0 also the pattern), x / y byte offsets
#define NSF2_YT_WORDS(m) ((m)->T * (((m)->nT + 2) * 4 + 1))   // per transform: the y offsets of every tile's groups, of rank 0
#define NSF2_LDS_BASE_FLOATS(m) (3 * (m)->Dp * 16 + 3 * (m)->Hp * 16 + 2 * NSF2_STAGE_FLOATS + 2 * NSF2_PART_FLOATS + 16 * 32 + \
                                 ((NSF2_TT_WORDS(m) + (m)->Dp + NSF2_YT_WORDS(m) + 3) & ~3))
// EAGER PARTIALS (round 4).  The burst wave's work for tile T1 grows with T1 (40 (T1 - 1) MFMAs at 32 cycles each against a
// chain that takes ~9 k cycles per tile whatever the tile): from the seventh tile on the chain waited for it (scripts/
// profile_nsf2_tiles.py: 0.5 / 2.1 / 3.1 k cycles at tiles 5 - 7 of a nine-tile flow) while on tiles 1 - 4 the burst wave waited
// 2 - 4 k cycles for the chain.  The output partials of the LAST TWO live tiles therefore start early: their ranks' products
// against h2 tiles 0, 1, 2 (last tile) and 0, 1 (the one before) are formed at steps 2, 3, 4 -- in the burst wave's idle
// time -- into two more partial buffers in LDS that only the burst wave touches; the two tiles' own steps start from those
// and run K = 3 .. / 2 .. only (nsf_burst_tile<T1, KS>).  Flows of >= 8 live tiles on the static path whose LDS stays within
// half a CU's (two workgroups per CU).
#define NSF2_EAGER_FLOATS (2 * NSF2_PART_FLOATS)
#define NSF2_EAGER_OK(m) ((m)->nT >= 8 && (m)->nT <= NSF2_PK + 1 && \
                          (size_t)(NSF2_LDS_BASE_FLOATS(m) + NSF2_EAGER_FLOATS) * sizeof(float) <= 80 * 1024)
#define NSF2_LDS_FLOATS(m) (NSF2_LDS_BASE_FLOATS(m) + (NSF2_EAGER_OK(m) ? NSF2_EAGER_FLOATS : 0))

__device__ __forceinline__ f32x4 as_acc(const float4& v) { f32x4 r; r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w; return r; }

// the chain's hidden-layer operands of a tile (static data, requested a tile ahead) and its table words
struct NsfHid {
    float4 wt1, wt2;               // diagonal tile of layers 1 / 2, rows transposed (chain_vo_T); component c = K chunk c
    float4 wn1, wn2;               // block (Tt + 1, Tt), transposed
    float4 w0o[4];                 // [group].jt: W0[row q of quad jt of this tile][the rank the group produces]
    float4 w0N[4];                 // the same against the next tile's quads
    int g[4], xy[4], yo[4], pat;   // ranks of the groups, byte offsets of their x word and of their y word in the input array (walker 0), quad pattern
};
// a rank's two output tiles against the previous (fp) and the own (fc) hidden tile, natural fragments
struct NsfOut { float4 fp0, fp1, fc0, fc1; };

struct NsfChain {
    float a0[4], p1[4], p2[4];
    f32x4 acc1, acc2, accN1, accN2;
    float a0N[4];
    float h0s[4], h1s[4], h2s[4];  // this tile's activations, row q of every quad
    float h2p[4];                  // the previous tile's h2 (B operands of the fp products)
    f32x4 o0, o1;                  // the current group's 23 parameters so far: staged partial + previous tile + the tile's earlier quads,
                                   // formed in the shadow of the PREVIOUS group's spline solve (round 4)
    RqsPend pend;                  // the previous group's solve, log-derivative not evaluated yet
    float pend_x;                  // its x: store and off-path rank-1 updates pending
};

// Groups I .. of a tile with quad pattern PAT, one after the other, straight-line.
//
// Round 4: ONLY WHAT WAITS FOR x STANDS BETWEEN TWO RANKS' x.  A rank's parameters are
//   staged partial (burst wave) + previous tile's h2 (8 MFMAs) + the own tile's earlier quads (2 per quad) + the group's own
//   quads (2 per quad, behind its hops),
// and only the last term depends on the x the previous group has just solved.  The first three used to stand at the head
// of the group -- ~10 MFMAs = 320 cycles of matrix pipe between x and the first hop that needs it -- while the spline solve
// before it (~170 vector instructions, two LDS round trips, no MFMA) left the pipe idle; and the solve's log-derivative,
// the x store and the rank-1 updates of the NEXT tile stood between x and the next hop although nothing there waits for them.
//   * The following group's MFMAs are issued in the numbered slots of this group's solve (rqs_inverse_split_sh: one MFMA
//     per slot, each behind >= 32 cycles of vector work or inside an LDS wait) into n0 / n1 -> s.o0 / s.o1:
//       - groups 1 .. NG-1: exactly the products and the order of additions they had (bit-identical);
//       - group 0 of the NEXT tile, in the last solve of this one: its previous tile is this tile, whose h2 is final by
//         then; the staged partial of the next tile only exists behind the barrier, so the accumulators start from zero
//         and the caller adds the partial at the head of the tile (a different order of additions for these ranks).
//   * What a solve leaves behind (s.pend: log-derivative, x store, the next tile's and the later quads' rank-1 updates)
//     runs in the next group's hops, between the MFMA the chain waits for and the right-looking one behind it -- 32 cycles
//     in which the wavefront could issue nothing anyway -- and at the end of the tile for its last group (nsf_settle).
// Fragments: group I reads ob[I & 1]; the requests keep their distance of one whole group: `ahead(G, slot)` with
// G = I + 1 asks for what follows group I + 1 -- a later group of this tile or the next tile's first -- into
// ob[(I + 2) & 1] = ob[I & 1], whose own-quad fragments the MFMAs just above were the last to read; G = NG (the next tile's
// first group is being prepared) asks for the group after THAT into ob[1].  A tile of an odd number of groups leaves the
// next tile's first group in ob[1]: its own-tile fragments (the previous-tile ones are used up here) move to ob[0].
template <int PAT, int J>
__device__ __forceinline__ void nsf_settle_a(NsfChain& s, const NsfHid& f, float* X, int q, int p, int keep_from) {
    // group J's x store and rank-1 updates off the dependent path: the next tile's quads, this tile's quads from `keep_from`
    constexpr int c1 = pat_end(PAT, J);
    const float xg = s.pend_x;
    if (q == 0) *reinterpret_cast<float*>(reinterpret_cast<char*>(X) + (p << 4) + f.xy[J]) = xg;
#pragma unroll
    for (int jt = c1 + 1; jt < 4; ++jt) if (jt >= keep_from) s.a0[jt] = fmaf(comp(f.w0o[J], jt), xg, s.a0[jt]);
    s.a0N[0] = fmaf(f.w0N[J].x, xg, s.a0N[0]); s.a0N[1] = fmaf(f.w0N[J].y, xg, s.a0N[1]);
    s.a0N[2] = fmaf(f.w0N[J].z, xg, s.a0N[2]); s.a0N[3] = fmaf(f.w0N[J].w, xg, s.a0N[3]);
}

template <int PAT, int I, class AH>
__device__ __forceinline__ void nsf_group(NsfChain& s, const NsfHid& f, NsfOut (&ob)[2], const float* part, float* X, const float* Y,
                                          float* PAR, int D, int q, int p, int lane, float& ladj, const AH& ahead,
                                          long long* pf = nullptr) {
#define NSF_STAMP(K) if (pf) { const long long now_ = clock64(); pf[K] += now_ - pf[15]; pf[15] = now_; }
    // a value nothing nearby waits for is computed HERE (instruction selection otherwise sinks a pure computation to its
    // first use -- the log-derivatives of a whole tile ended up behind its last group, on the dependent path)
#define NSF_PIN(V) asm volatile("" : "+v"(V));
    constexpr int NG = pat_ngroups(PAT);
    if constexpr (I < NG) {
        constexpr int c0 = pat_start(PAT, I), c1 = pat_end(PAT, I);
        {
            const NsfOut& o = ob[I & 1];
            std::integral_constant<int, I> gi;
            std::integral_constant<int, I + 1> gn;
            std::integral_constant<int, NG> ngc;
            NSF_STAMP(7)
            float h0[4], h1[4], h2[4];
            // ---------------------------------------------------------------- hop 1
#pragma unroll
            for (int c = c0; c <= c1; ++c) { h0[c] = fmaxf(s.a0[c], 0.0f); s.h0s[c] = h0[c]; }
            CHAIN_FENCE();
#pragma unroll
            for (int c = c0; c <= c1; ++c) s.acc1 = MFMA(comp(f.wt1, c), h0[c], s.acc1);
            CHAIN_FENCE();
            if constexpr (I > 0) nsf_settle_a<PAT, I - 1>(s, f, X, q, p, c1 + 1);      // (the previous group's side effects)
            CHAIN_FENCE();
#pragma unroll
            for (int c = c0; c <= c1; ++c) s.accN1 = MFMA(comp(f.wn1, c), h0[c], s.accN1);
            CHAIN_FENCE();
            ahead(gi, std::integral_constant<int, 4>{}, ngc);
            CHAIN_FENCE();
#pragma unroll
            for (int c = c0; c <= c1; ++c) { h1[c] = fmaxf((s.acc1[c] + s.p1[c]) + h0[c], 0.0f); s.h1s[c] = h1[c]; }
            CHAIN_FENCE();
            // ---------------------------------------------------------------- hop 2
#pragma unroll
            for (int c = c0; c <= c1; ++c) s.acc2 = MFMA(comp(f.wt2, c), h1[c], s.acc2);
            CHAIN_FENCE();
            if constexpr (I > 0) { rqs_ladj_1(s.pend); NSF_PIN(s.pend.rden) }
            // (this group's y, and the staged partial the following group's parameters start from)
            const float yv = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(Y) + (p << 4) + f.yo[I]);
            constexpr bool LAST = I + 1 >= NG;
            f32x4 n0 = f32x4{0.f, 0.f, 0.f, 0.f}, n1 = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (!LAST) {
                n0 = as_acc(*reinterpret_cast<const float4*>(part + (2 * (I + 1)) * 256 + (lane << 2)));
                n1 = as_acc(*reinterpret_cast<const float4*>(part + (2 * (I + 1) + 1) * 256 + (lane << 2)));
            }
            CHAIN_FENCE();
#pragma unroll
            for (int c = c0; c <= c1; ++c) s.accN2 = MFMA(comp(f.wn2, c), h1[c], s.accN2);
            CHAIN_FENCE();
            ahead(gi, std::integral_constant<int, 5>{}, ngc);
            CHAIN_FENCE();
#pragma unroll
            for (int c = c0; c <= c1; ++c) { h2[c] = fmaxf((s.acc2[c] + s.p2[c]) + h1[c], 0.0f); s.h2s[c] = h2[c]; }
            CHAIN_FENCE();
            NSF_STAMP(2)
            // ---------------------------------------------------------------- hop 3: the group's own quads
            if (!(NSF2_ABL & 2)) {
#pragma unroll
                for (int c = c0; c <= c1; ++c) s.o0 = MFMA(comp(o.fc0, c), h2[c], s.o0);
                CHAIN_FENCE();
                if constexpr (I > 0) { rqs_ladj_2(s.pend); NSF_PIN(s.pend.jac) }
                CHAIN_FENCE();
#pragma unroll
                for (int c = c0; c <= c1; ++c) s.o1 = MFMA(comp(o.fc1, c), h2[c], s.o1);
            }
            CHAIN_FENCE();
            NSF_STAMP(3)
            // ---------------------------------------------------------------- this rank's spline solve; in its slots the
            // following group's parameters as far as they do not wait for that group's hops
            constexpr int d0 = LAST ? 0 : pat_start(PAT, I + 1);
            constexpr int NM = 8 + 2 * d0;
            static_assert(NM <= RQS_NSLOTS, "more shadow MFMAs than slots");
            const NsfOut& on = ob[(I + 1) & 1];
            const float* hB = LAST ? s.h2s : s.h2p;                       // the following group's previous tile
            float xg;
            RqsPend pn;
            auto shadow = [&](auto k_) {
                constexpr int K = decltype(k_)::value;
                if constexpr (K < 8) {
                    constexpr int j = K >> 1;
                    if constexpr ((K & 1) == 0) { if (!(NSF2_ABL & 2)) n0 = MFMA(comp(on.fp0, j), hB[j], n0); }
                    else {
                        if (!(NSF2_ABL & 2)) n1 = MFMA(comp(on.fp1, j), hB[j], n1);
                        if constexpr (!(LAST && (NG & 1))) ahead(gn, std::integral_constant<int, j>{}, ngc);
                        else if constexpr (K == 7) {                        // (the requests' target is the buffer the MFMAs read)
                            ob[0].fc0 = ob[1].fc0; ob[0].fc1 = ob[1].fc1;
                            static_for<4>([&](auto sl_) { ahead(gn, sl_, ngc); });
                        }
                    }
                } else if constexpr (K < NM) {
                    constexpr int c = (K - 8) >> 1;
                    if (!(NSF2_ABL & 2)) {
                        if constexpr ((K & 1) == 0) n0 = MFMA(comp(on.fc0, c), s.h2s[c], n0);
                        else n1 = MFMA(comp(on.fc1, c), s.h2s[c], n1);
                    }
                }
                if constexpr (K == 5 && I > 0) { ladj -= rqs_ladj_3(s.pend); NSF_PIN(ladj) }          // (inside the exchange's LDS wait)
            };
            if (NSF2_ABL & 1) {
                float* pr = PAR + (p << 5) + (q << 2);
                *reinterpret_cast<float4*>(pr) = make_float4(s.o0[0], s.o0[1], s.o0[2], s.o0[3]);
                *reinterpret_cast<float4*>(pr + 16) = make_float4(s.o1[0], s.o1[1], s.o1[2], s.o1[3]);
                WAVE_LDS_FENCE();
                xg = yv + PAR[(p << 5)] + PAR[(p << 5) + 22];
                pn.s = 1.0f; pn.e = 0.0f; pn.z = PAR[(p << 5) + 8]; pn.d0 = pn.d1 = 1.0f; pn.inside = true;
                static_for<RQS_NSLOTS>(shadow);
            } else {
                rqs_inverse_split_sh(s.o0, s.o1, PAR + (p << 5), q, yv, xg, pn, shadow);
            }
            CHAIN_FENCE();
            NSF_STAMP(4)
            // ---------------------------------------------------------------- what the next hop waits for: the rank-1 update of the
            // following group's quads; everything else of this solve stays pending
            if constexpr (!LAST) {
                constexpr int e1 = pat_end(PAT, I + 1);
#pragma unroll
                for (int jt = c1 + 1; jt <= e1; ++jt) s.a0[jt] = fmaf(comp(f.w0o[I], jt), xg, s.a0[jt]);
            }
            s.pend = pn;
            s.pend_x = xg;
            s.o0 = n0; s.o1 = n1;
            CHAIN_FENCE();
            NSF_STAMP(5)
            nsf_group<PAT, I + 1, AH>(s, f, ob, part, X, Y, PAR, D, q, p, lane, ladj, ahead, pf);
        }
    } else {
        // the last group's side effects
        nsf_settle_a<PAT, NG - 1>(s, f, X, q, p, 4);
        rqs_ladj_1(s.pend); rqs_ladj_2(s.pend);
        ladj -= rqs_ladj_3(s.pend);
        WAVE_LDS_FENCE();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// BURST wave, one tile (flows of up to NSF2_PK + 1 live hidden tiles: every count below is a compile-time constant, the
// tile index is dispatched once per tile).  A lone wavefront issues in order: a block of loads costs ~60 cycles apiece and
// a block of MFMAs 32 apiece, but loads issued BETWEEN MFMAs cost nothing (the matrix pipe is busy anyway), and nothing is
// requested that is not used.  So the next tile's hidden operands are requested in the shadows of this tile's hidden
// products, and rank r + 1's output fragments in the shadows of rank r's output products.
struct NsfBurstSet { float4 p1[NSF2_PK], p2[NSF2_PK], xf[NSF2_PX], b0, b1, b2; };       // (the streamed path's operand sets)
// what crosses a tile boundary on the burst wave: the next tile's fragment sets of its first two K steps and its biases --
// a set = the four ranks' first output tiles (widths | heights) + the two rank PAIRS' shared derivative tiles
struct NsfBurstCarry { float4 f[2][6], b[6]; };
struct NsfBurstCtx {
    __amdgpu_buffer_rsrc_t rs;
    int tb, oF1, oF2, oF0C, oB0T, oB1T, oB2T, oF3I, oB3I, nT, nXT, D, lane, vo_lane, vo_T, vo_q;
    const float *X, *H0, *H1, *H2;
    float *stg, *part;
    int g[4], gnx[4], ksn;         // the ranks of this tile's groups; of the next tile's groups, and the first K step of its own products
    long long* ts;                 // (measurement only, NSF2_TILE_STAMPS: where this tile's section stamps go, or null)
    int eag;                       // eager partials on: eg[0] / ea[0]: the ranks / the partial buffer of the last live tile, [1]: of the one before
    int eg[2][4];
    float* ea[2];
};

// THE OUTPUT PARTIALS, K-OUTER (round 4).  A rank's 23 parameters are two 16-row output tiles, the second one carrying 7
// rows (the derivatives): per rank and K tile the burst wave spent 8 MFMAs, 2 fragment loads and one LDS read of h2 on 23
// useful rows of 32, and it is at the matrix pipe's limit on the late tiles.  The derivative rows of TWO ranks share one
// tile now -- rows 0-7 from the pair's first rank, rows 8-15 from its second: the same fragment records, addressed per
// lane (the second rank's rows through an offset in the lane's VGPR; a padding rank's lanes out of range: zeros) --
// so a tile's four ranks are 6 accumulators instead of 8, and the loop runs K tiles OUTERMOST with all six resident:
// per K step one read of h2, 6 fragment loads and 24 MFMAs (was 4 x (1 + 2 + 8)), each accumulator's products in the
// order they had (x, y, z, w of K ascending: the same bits).  The fragments of step j + 2 are requested in the shadows of
// step j's MFMAs into a ring of three sets; the first two sets and the biases come with the previous tile (carry).  The
// staged partials keep their layout: a pair's shared tile is stored twice, the second time with the wavefront's halves
// swapped, so that the chain finds the second rank's derivative rows where it always read them.
template <int T1, int KS = 0, bool EAG = false>
__device__ __forceinline__ void nsf_burst_tile(const NsfBurstCtx& c, NsfBurstCarry& carry) {
    constexpr int NK = T1 > 0 ? T1 - 1 : 0;          // final hidden tiles: 0 .. T1-2
    constexpr int NF = NK > 0 ? NK : 1;
    constexpr int SH = NK - KS;                      // K steps of this tile's own products
    static_assert(T1 <= NSF2_PK && KS <= NK, "tile beyond the static path");
    const int lane = c.lane, q = lane >> 4;
    float4 hp1[NF], hp2[NF], xf[NSF2_PX], hb0, hb1, hb2;
    const int stride = 2 * c.nT * 1024;              // bytes between two ranks' fragment blocks
    auto obase = [&](const int tb, const int g) { return tb + c.oF3I + (g < c.D ? g : 0) * stride; };
    auto ovo = [&](const int g) { return g < c.D ? c.vo_lane : NSF2_OOB; };
    // lane offset of a pair's shared derivative fragment (rows 0-7: rank ga, rows 8-15: rank gb), relative to ga's block
    auto pair_vo = [&](const int ga, const int gb) {
        const int i = lane & 15, base = ((q << 4) + (i & 7)) << 4;
        return i < 8 ? (ga < c.D ? base : NSF2_OOB) : (gb < c.D ? base + (gb - ga) * stride : NSF2_OOB);
    };
    // ... and of its shared bias rows (lanes q < 2: ga's derivative rows 4q .., lanes q >= 2: gb's rows 4 (q - 2) ..)
    auto pair_bvo = [&](const int ga, const int gb) {
        return q < 2 ? (ga < c.D ? (q << 4) : NSF2_OOB) : (gb < c.D ? ((q - 2) << 4) + (gb - ga) * 128 : NSF2_OOB);
    };
    // bias k of a tile (0-3: the ranks' first halves, 4 / 5: the pairs' shared second halves)
    auto bias = [&](const int tb, const int (&g)[4], const int bv0, const int bv1, const int k) {
        if (k < 4) return bload4(c.rs, g[k] < c.D ? c.vo_q : NSF2_OOB, tb + c.oB3I + (g[k] < c.D ? g[k] : 0) * 128);
        const int ga = g[k == 4 ? 0 : 2];
        return bload4(c.rs, k == 4 ? bv0 : bv1, tb + c.oB3I + (ga < c.D ? ga : 0) * 128 + 64);
    };
    const int pv0 = pair_vo(c.g[0], c.g[1]), pv1 = pair_vo(c.g[2], c.g[3]);
    int fb_[6], fvo_[6];                              // this tile's fragment blocks: K tile 0 of fragment k of a set
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int g = c.g[k < 4 ? k : (k == 4 ? 0 : 2)];
        fb_[k] = obase(c.tb, g) + (k < 4 ? 0 : c.nT) * 1024;
        fvo_[k] = k < 4 ? ovo(g) : (k == 4 ? pv0 : pv1);
    }
    // ---- (0) eager partials of the last two live tiles (EAG; steps 2, 3, 4): this step's jobs are job 0 = the last tile's
    // ranks against h2 tile T1 - 2 and, from step 3, job 1 = the ranks of the tile before against h2 tile T1 - 3; step 2 is the
    // first touch of the last tile's partials, step 3 of the other tile's (they start from the ranks' biases).  Their
    // fragments (and biases) are requested in the shadows of the regular products' MFMAs, `eload(k)`, k = 0 .. 4 EPR - 1,
    // and multiplied at the end of the tile.  (The eager partials keep one pair of tiles per rank.)
    constexpr int NJ = !EAG ? 0 : (T1 == 2 ? 1 : ((T1 == 3 || T1 == 4) ? 2 : 0));
    static_assert(!EAG || NJ > 0, "eager work exists at steps 2, 3, 4 only");
    constexpr int NJ1 = NJ > 0 ? NJ : 1;
    constexpr int JK0 = T1 - 2, JK1 = T1 - 3;
    constexpr bool FIRST0 = T1 == 2, FIRST1 = T1 == 3;
    constexpr int EPR = 2 * NJ + ((FIRST0 || FIRST1) ? 2 : 0);        // requests per rank
    float4 ef0[NJ1][4], ef1[NJ1][4], eb0[4], eb1[4];
    auto eload = [&](const int k) {
        if constexpr (NJ > 0) {
            if (k >= 4 * EPR) return;
            const int r = k / EPR, w = k - r * EPR;
            const int fb = FIRST0 ? 0 : 1;                           // the tile whose partials are touched first at this step
            if (w < 2 * NJ) {
                const int j = w >> 1, g = c.eg[j][r], K = j == 0 ? JK0 : JK1;
                if ((w & 1) == 0) ef0[j][r] = bload4(c.rs, ovo(g), obase(c.tb, g) + K * 1024);
                else ef1[j][r] = bload4(c.rs, ovo(g), obase(c.tb, g) + (c.nT + K) * 1024);
            } else {
                const int g = c.eg[fb][r];
                const int so = c.tb + c.oB3I + (g < c.D ? g : 0) * 128, vo = g < c.D ? c.vo_q : NSF2_OOB;
                if (w == 2 * NJ) eb0[r] = bload4(c.rs, vo, so);
                else eb1[r] = bload4(c.rs, vo, so + 64);
            }
        }
    };
    const int soH1 = c.tb + c.oF1 + T1 * c.nT * 1024, soH2 = c.tb + c.oF2 + T1 * c.nT * 1024;
    // the tile's other requests, in the shadows of the K steps (or as a block, SH == 0): the hidden layers' fragments and
    // biases, the layer-0 fragments, the eager jobs' operands
    constexpr int NXL = 2 * NK + NSF2_PX + 3 + 4 * EPR;
    auto xload = [&](const int k) {
        if (k < NK) hp1[k] = bload4(c.rs, c.vo_T, soH1 + k * 1024);
        else if (k < 2 * NK) hp2[k - NK] = bload4(c.rs, c.vo_T, soH2 + (k - NK) * 1024);
        else if (k < 2 * NK + NSF2_PX) { const int i = k - 2 * NK; xf[i] = bload4(c.rs, i < c.nXT ? c.vo_T : NSF2_OOB, c.tb + c.oF0C + (T1 * c.nXT + i) * 1024); }
        else if (k == 2 * NK + NSF2_PX) hb0 = bload4(c.rs, c.vo_q, c.tb + c.oB0T + 64 * T1);
        else if (k == 2 * NK + NSF2_PX + 1) hb1 = bload4(c.rs, c.vo_q, c.tb + c.oB1T + 64 * T1);
        else if (k == 2 * NK + NSF2_PX + 2) hb2 = bload4(c.rs, c.vo_q, c.tb + c.oB2T + 64 * T1);
        else if (k < NXL) eload(k - (2 * NK + NSF2_PX + 3));
    };
#define NSF_BSTAMP(K) if (c.ts && lane == 0) c.ts[K] = clock64();
    NSF_BSTAMP(0)
    // ---- (1) output partials: six accumulators, K tiles outermost
    f32x4 o0[4], o1c[2];
    if constexpr (KS > 0) {                                            // an eager tile starts from its partials
        const float* ea = c.ea[KS == 3 ? 0 : 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) o0[r] = as_acc(*reinterpret_cast<const float4*>(ea + (2 * r) * 256 + (lane << 2)));
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int r = 2 * pr + (q >> 1), l2 = q < 2 ? lane : lane - 32;          // (the pair's second rank: its lanes q = 0, 1)
            o1c[pr] = as_acc(*reinterpret_cast<const float4*>(ea + (2 * r + 1) * 256 + (l2 << 2)));
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) o0[r] = as_acc(carry.b[r]);
        o1c[0] = as_acc(carry.b[4]); o1c[1] = as_acc(carry.b[5]);
    }
    float4 S[3][6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { S[0][k] = carry.f[0][k]; S[1][k] = carry.f[1][k]; }
    if constexpr (SH == 0) {
#pragma unroll
        for (int k = 0; k < NXL; ++k) xload(k);
    }
    float4 bnx = *reinterpret_cast<const float4*>(c.H2 + (KS << 8) + (lane << 2));
    static_for<SH>([&](auto j_) {
        constexpr int j = decltype(j_)::value, i = KS + j, cur = j % 3, nxt = (j + 2) % 3;
        constexpr int SH1 = SH > 0 ? SH : 1;
        constexpr int XPS = (NXL + SH1 - 1) / SH1;                     // other requests per step
        const float4 b = bnx;
        if constexpr (i + 1 < NK) bnx = *reinterpret_cast<const float4*>(c.H2 + ((i + 1) << 8) + (lane << 2));
        auto ld = [&](const int k) {
            if constexpr (j + 2 < SH) { if (!(NSF2_ABL & 8)) S[nxt][k] = bload4(c.rs, fvo_[k], fb_[k] + (i + 2) * 1024); }
        };
        auto xs = [&](const int part) {                                // this step's share of the other requests, in four parts
#pragma unroll
            for (int l = 0; l < XPS; ++l) if ((l & 3) == part) xload(j * XPS + l);
        };
        if (!(NSF2_ABL & 4)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o0[r] = MFMA(S[cur][r].x, b.x, o0[r]);
            o1c[0] = MFMA(S[cur][4].x, b.x, o1c[0]); o1c[1] = MFMA(S[cur][5].x, b.x, o1c[1]);
        }
        ld(0); ld(1); xs(0);
        if (!(NSF2_ABL & 4)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o0[r] = MFMA(S[cur][r].y, b.y, o0[r]);
            o1c[0] = MFMA(S[cur][4].y, b.y, o1c[0]); o1c[1] = MFMA(S[cur][5].y, b.y, o1c[1]);
        }
        ld(2); ld(3); xs(1);
        if (!(NSF2_ABL & 4)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o0[r] = MFMA(S[cur][r].z, b.z, o0[r]);
            o1c[0] = MFMA(S[cur][4].z, b.z, o1c[0]); o1c[1] = MFMA(S[cur][5].z, b.z, o1c[1]);
        }
        ld(4); ld(5); xs(2);
        if (!(NSF2_ABL & 4)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o0[r] = MFMA(S[cur][r].w, b.w, o0[r]);
            o1c[0] = MFMA(S[cur][4].w, b.w, o1c[0]); o1c[1] = MFMA(S[cur][5].w, b.w, o1c[1]);
        }
        xs(3);
        CHAIN_FENCE();
    });
#pragma unroll
    for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float4*>(c.part + (2 * r) * 256 + (lane << 2)) = make_float4(o0[r][0], o0[r][1], o0[r][2], o0[r][3]);
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const float4 v = make_float4(o1c[pr][0], o1c[pr][1], o1c[pr][2], o1c[pr][3]);
        *reinterpret_cast<float4*>(c.part + (4 * pr + 1) * 256 + (lane << 2)) = v;                 // the pair's first rank
        *reinterpret_cast<float4*>(c.part + (4 * pr + 3) * 256 + ((lane ^ 32) << 2)) = v;          // its second: halves swapped
    }
    NSF_BSTAMP(1)
    // ---- (2) hidden layers against the final tiles; the next tile's first two fragment sets and its biases in the shadows
    f32x4 a0 = as_acc(hb0), a1 = as_acc(hb1), a2 = as_acc(hb2);
    const int npv0 = pair_vo(c.gnx[0], c.gnx[1]), npv1 = pair_vo(c.gnx[2], c.gnx[3]);
    const int nbv0 = pair_bvo(c.gnx[0], c.gnx[1]), nbv1 = pair_bvo(c.gnx[2], c.gnx[3]);
    // the carry: the sets of the next tile's first NCS K steps (it has T1 of them less those its eager steps took) + six biases;
    // block bases once per tile (a request is then one scalar add)
    constexpr int NCS = T1 >= 2 ? 2 : T1, NCL = 6 * NCS + 6;
    int nb_[6], nvo_[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int g = c.gnx[k < 4 ? k : (k == 4 ? 0 : 2)];
        nb_[k] = obase(c.tb, g) + ((k < 4 ? 0 : c.nT) + c.ksn) * 1024;
        nvo_[k] = k < 4 ? ovo(g) : (k == 4 ? npv0 : npv1);
    }
    auto cload = [&](const int k) {                                    // request k of NCL of the carry
        if (NSF2_ABL & 8) return;
        if (k < 6 * NCS) carry.f[k / 6][k % 6] = bload4(c.rs, nvo_[k % 6], nb_[k % 6] + (k / 6) * 1024);
        else if (k < NCL) carry.b[k - 6 * NCS] = bias(c.tb, c.gnx, nbv0, nbv1, k - 6 * NCS);
    };
    float4 b1n = *reinterpret_cast<const float4*>(c.H0 + (lane << 2)), b2n = *reinterpret_cast<const float4*>(c.H1 + (lane << 2));
    constexpr int NK1 = NK > 0 ? NK : 1;
    constexpr int CPS = (NCL + NK1 - 1) / NK1;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const float4 b1 = b1n, b2 = b2n;
        if (i + 1 < NK) {
            b1n = *reinterpret_cast<const float4*>(c.H0 + ((i + 1) << 8) + (lane << 2));
            b2n = *reinterpret_cast<const float4*>(c.H1 + ((i + 1) << 8) + (lane << 2));
        }
        if (!(NSF2_ABL & 32)) {
            a1 = MFMA(hp1[i].x, b1.x, a1); a2 = MFMA(hp2[i].x, b2.x, a2);
            a1 = MFMA(hp1[i].y, b1.y, a1); a2 = MFMA(hp2[i].y, b2.y, a2);
        }
#pragma unroll
        for (int l = 0; l < CPS; ++l) if ((l & 1) == 0) cload(i * CPS + l);
        if (!(NSF2_ABL & 32)) {
            a1 = MFMA(hp1[i].z, b1.z, a1); a2 = MFMA(hp2[i].z, b2.z, a2);
            a1 = MFMA(hp1[i].w, b1.w, a1); a2 = MFMA(hp2[i].w, b2.w, a2);
        }
#pragma unroll
        for (int l = 0; l < CPS; ++l) if ((l & 1) == 1) cload(i * CPS + l);
        CHAIN_FENCE();
    }
    if constexpr (NK == 0) {
#pragma unroll
        for (int k = 0; k < NCL; ++k) cload(k);
    }
    NSF_BSTAMP(2)
    // layer 0 against the ranks of tiles <= T1-2 (the chain adds the ranks of tile T1-1 itself)
#pragma unroll
    for (int i = 0; i < NSF2_PX; ++i) {
        if (i < c.nXT) {
            const float4 b = *reinterpret_cast<const float4*>(c.X + (i << 8) + (lane << 2));
            a0 = MFMA(xf[i].x, b.x, a0); a0 = MFMA(xf[i].y, b.y, a0);
            a0 = MFMA(xf[i].z, b.z, a0); a0 = MFMA(xf[i].w, b.w, a0);
        }
    }
    *reinterpret_cast<float4*>(c.stg + (lane << 2)) = make_float4(a0[0], a0[1], a0[2], a0[3]);
    *reinterpret_cast<float4*>(c.stg + 256 + (lane << 2)) = make_float4(a1[0], a1[1], a1[2], a1[3]);
    *reinterpret_cast<float4*>(c.stg + 512 + (lane << 2)) = make_float4(a2[0], a2[1], a2[2], a2[3]);
    NSF_BSTAMP(3)
    // ---- (3) this step's eager jobs: read the partials (all of them up front: one LDS latency), multiply, write back
    if constexpr (NJ > 0) {
        const float4 hk0 = *reinterpret_cast<const float4*>(c.H2 + (JK0 << 8) + (lane << 2));
        float4 hk1 = hk0;
        if constexpr (NJ > 1) hk1 = *reinterpret_cast<const float4*>(c.H2 + (JK1 << 8) + (lane << 2));
        f32x4 o0[NJ1][4], o1[NJ1][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bool first = (j == 0 && FIRST0) || (j == 1 && FIRST1);
                const float* ap = c.ea[j] + (2 * r) * 256 + (lane << 2);
                if (first) { o0[j][r] = as_acc(eb0[r]); o1[j][r] = as_acc(eb1[r]); }
                else { o0[j][r] = as_acc(*reinterpret_cast<const float4*>(ap)); o1[j][r] = as_acc(*reinterpret_cast<const float4*>(ap + 256)); }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float4 b = j == 0 ? hk0 : hk1;
                const float4 F0 = ef0[j][r], F1 = ef1[j][r];
                o0[j][r] = MFMA(F0.x, b.x, o0[j][r]); o1[j][r] = MFMA(F1.x, b.x, o1[j][r]);
                o0[j][r] = MFMA(F0.y, b.y, o0[j][r]); o1[j][r] = MFMA(F1.y, b.y, o1[j][r]);
                o0[j][r] = MFMA(F0.z, b.z, o0[j][r]); o1[j][r] = MFMA(F1.z, b.z, o1[j][r]);
                o0[j][r] = MFMA(F0.w, b.w, o0[j][r]); o1[j][r] = MFMA(F1.w, b.w, o1[j][r]);
                float* ap = c.ea[j] + (2 * r) * 256 + (lane << 2);
                *reinterpret_cast<float4*>(ap) = make_float4(o0[j][r][0], o0[j][r][1], o0[j][r][2], o0[j][r][3]);
                *reinterpret_cast<float4*>(ap + 256) = make_float4(o1[j][r][0], o1[j][r][1], o1[j][r][2], o1[j][r][3]);
            }
        }
    }
}

// Two accumulators over K tiles K0 .. nK-1 with the fragments of four K tiles in flight (the streamed path of the wide
// flows): acc0 += W0[K] . act0[K], acc1 += W1[K] . act1[K]; fragment K of operand j at byte offset soj + K * 1024.
__device__ __forceinline__ void nsf_stream2(f32x4& acc0, f32x4& acc1, __amdgpu_buffer_rsrc_t rs, const int vo, const int so0, const int so1,
                                            const int K0, const int nK, const float* act0, const float* act1, const int lane) {
    if (K0 >= nK) return;
    float4 w0r[4], w1r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int v = K0 + j < nK ? vo : NSF2_OOB;
        w0r[j] = bload4(rs, v, so0 + (K0 + j) * 1024);
        w1r[j] = bload4(rs, v, so1 + (K0 + j) * 1024);
    }
    for (int Kb = K0; Kb < nK; Kb += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int K = Kb + j;
            const float4 w0 = w0r[j], w1 = w1r[j];
            const int v = K + 4 < nK ? vo : NSF2_OOB;
            w0r[j] = bload4(rs, v, so0 + (K + 4) * 1024);
            w1r[j] = bload4(rs, v, so1 + (K + 4) * 1024);
            if (K < nK) {
                const float4 b0 = *reinterpret_cast<const float4*>(act0 + (K << 8) + (lane << 2));
                const float4 b1 = *reinterpret_cast<const float4*>(act1 + (K << 8) + (lane << 2));
                acc0 = MFMA(w0.x, b0.x, acc0); acc1 = MFMA(w1.x, b1.x, acc1);
                acc0 = MFMA(w0.y, b0.y, acc0); acc1 = MFMA(w1.y, b1.y, acc1);
                acc0 = MFMA(w0.z, b0.z, acc0); acc1 = MFMA(w1.z, b1.z, acc1);
                acc0 = MFMA(w0.w, b0.w, acc0); acc1 = MFMA(w1.w, b1.w, acc1);
            }
        }
    }
}

// flows with more live hidden tiles than the static burst tile covers: a rank's output partial, fragments streamed
__device__ __forceinline__ void nsf_out_partials_wide(__amdgpu_buffer_rsrc_t rs, const int tb_f3i, const int tb_b3i, const int nT, const int D,
                                                      const int nK, const int g, const float* H2, float* d, const int lane,
                                                      const int vo_lane, const int vo_q) {
    const bool lv = g < D;
    const int gg = lv ? g : 0;
    f32x4 o0 = as_acc(bload4(rs, lv ? vo_q : NSF2_OOB, tb_b3i + gg * 128)), o1 = as_acc(bload4(rs, lv ? vo_q : NSF2_OOB, tb_b3i + gg * 128 + 64));
    if (lv) nsf_stream2(o0, o1, rs, vo_lane, tb_f3i + gg * 2 * nT * 1024, tb_f3i + (gg * 2 + 1) * nT * 1024, 0, nK, H2, H2, lane);
    *reinterpret_cast<float4*>(d + (lane << 2)) = make_float4(o0[0], o0[1], o0[2], o0[3]);
    *reinterpret_cast<float4*>(d + 256 + (lane << 2)) = make_float4(o1[0], o1[1], o1[2], o1[3]);
}

// FM: 0 = plain inverse of `in`; 4 / 8 / 16 = fused proposal (pocomc/mcmc.py:77-85) of the workgroup's 16 walkers as the
// prologue (D <= 4 FM) and, when pa.epi.on, the scaler (+ prior) on them as the epilogue (scaler_body.h); -1: the plain
// inverse with cycle stamps (measurement only)
template <int FM>
__global__ __launch_bounds__(128) void maf_inverse_nsf2_kernel(pmc_maf_t m, const float* __restrict__ in, float* __restrict__ out,
                                                               float* __restrict__ ladj_out, int64_t n, long long* prof, ProposeArgs pa) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, p = lane & 15;
    const int D = m.D, Dp = m.Dp, Hp = m.Hp, T = m.T, nT = m.nT, nXT = m.nXT, nOT = m.nOT;
    const int nTl = __builtin_amdgcn_readfirstlane(m.meta[7]);
    // (measurement only, FM == -1: cycles of the chain wave of workgroup 0 by section, summed over the sweep)
    long long pfv[16];
#ifdef NSF2_TILE_STAMPS
    long long* pf = nullptr;
#else
    long long* pf = (FM == -1 && prof && blockIdx.x == 0) ? pfv : nullptr;
#endif
    if (FM == -1) for (int i = 0; i < 16; ++i) pfv[i] = 0;
    const int64_t row0 = (int64_t)blockIdx.x * 16;
    float* Y = smem;
    float* XA = Y + Dp * 16;
    float* XB = XA + Dp * 16;
    float* H0 = XB + Dp * 16;
    float* H1 = H0 + Hp * 16;
    float* H2 = H1 + Hp * 16;
    float* STG = H2 + Hp * 16;                 // two hidden staging buffers (tile parity)
    float* PART = STG + 2 * NSF2_STAGE_FLOATS; // two output staging buffers (tile parity)
    float* PAR = PART + 2 * NSF2_PART_FLOATS;  // [16 rows][32]: the exchange panel of the current rank (rqs_inverse_split)
    int* DGT = reinterpret_cast<int*>(PAR + 16 * 32);
    int* PRM = DGT + NSF2_TT_WORDS(&m);
    int* YT = PRM + Dp;
    int* Y0T = YT + T * (nT + 2) * 4;
    float* EAG = smem + NSF2_LDS_BASE_FLOATS(&m);      // (only with NSF2_EAGER_OK: the launcher sized the block by the same macro)
    const int* feat_of_rank = m.meta + 8;
    const int* rank_of_feat = m.meta + 8 + T * D;
    const int* quad_meta = m.meta + 8 + 2 * T * D;

    // byte offsets of the packed arrays inside one transform's block (maf_spec.py: pk_offsets of the spline image)
    const int oF1 = nT * nXT * 1024;
    const int oF2 = oF1 + nT * nT * 1024;
    const int oF3 = oF2 + nT * nT * 1024;
    const int oW0 = oF3 + nOT * nT * 1024;
    const int oB0 = oW0 + Dp * Hp * 4;
    const int oB3 = oB0 + 3 * Hp * 4;
    const int oF3I = oB3 + nOT * 64;
    const int oB3I = oF3I + D * 2 * nT * 1024;
    const int oCW0 = oB3I + D * 128;
    const int oF0C = oCW0 + nT * 1024;
    const int oB0T = oF0C + nT * nXT * 1024;
    const int oB1T = oB0T + Hp * 4;
    const int oB2T = oB1T + Hp * 4;
    const int blk_bytes = (int)(m.pk_per_transform * 4);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)m.packed, 0, blk_bytes * T, 0x00020000);
    const int vo_lane = lane << 4;
    const int vo_T = chain_vo_T(lane);
    const int vo_q = q << 4;
    auto lds_bar = []() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // (measurement only, a library built with -DNSF2_TILE_STAMPS, FM == -1: when the two wavefronts of workgroup 0 reach and
    //  leave every barrier -- prof[16 + 4 * barrier + {0, 1} chain, {2, 3} burst]; scripts/profile_nsf2_tiles.py)
#ifdef NSF2_TILE_STAMPS
    long long* ts = (FM == -1 && prof && blockIdx.x == 0) ? prof + 16 : nullptr;
    int bar_no = 0;
    auto lds_bar_t = [&]() {
        if (ts && lane == 0) ts[4 * bar_no + 2 * wv] = clock64();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (ts && lane == 0) ts[4 * bar_no + 2 * wv + 1] = clock64();
        ++bar_no;
    };
#else
    auto lds_bar_t = lds_bar;
#endif

    // per hidden tile 8 words (two rows of "no groups" behind the last tile): the ranks its groups produce (word 0 also
    // the quad pattern << 16), the byte offsets of those ranks' x / y word (walker 0)
    for (int e = threadIdx.x; e < (nT + 2) * 8; e += 128) {
        const int tile = e >> 3, k = e & 7, i = k & 3;
        int g = D, pat = 1;
        if (tile < nT) {
            int4 dg = *reinterpret_cast<const int4*>(quad_meta + 4 * tile);
            dg.x &= 0xffff; dg.y &= 0xffff; dg.z &= 0xffff; dg.w &= 0xffff;
            const bool ny = dg.y != dg.x, nz = dg.z != dg.y, nw = dg.w != dg.z;
            pat = 1 | (ny << 1) | (nz << 2) | (nw << 3);
            const int g1 = ny ? dg.y : (nz ? dg.z : (nw ? dg.w : D));
            const int g2 = ny ? (nz ? dg.z : (nw ? dg.w : D)) : ((nz && nw) ? dg.w : D);
            const int g3 = (ny && nz && nw) ? dg.w : D;
            g = i == 0 ? dg.x : (i == 1 ? g1 : (i == 2 ? g2 : g3));
            // padding groups (degree >= D: zero weights, zero bias) trail; the last live group absorbs their quads -- the
            // chain then runs live groups only and needs no per-group condition (a condition around a group makes every
            // register it updates a conditional assignment: ~200 copies per group)
            const int ng = 1 + ny + nz + nw;
            const int gs[4] = {dg.x, g1, g2, g3};
            int keep = 0, seen = 0;
            for (int j = 0; j < 4; ++j) {
                if ((pat >> j) & 1) { if (seen < ng && gs[seen] < D) keep |= 1 << j; ++seen; }
            }
            pat = keep ? keep : 1;
        }
        const int gg = g < D ? g : 0;
        DGT[e] = k < 4 ? (g | (i == 0 ? pat << 16 : 0)) : 4 * (((gg >> 4) << 8) + ((gg & 3) << 6) + ((gg >> 2) & 3));
    }
    for (int r = threadIdx.x; r < D; r += 128) PRM[r] = feat_of_rank[r];     // (the last transform's x is stored by feature)
    // No re-ranking between transforms: transform t reads its input y where the transform before it (t + 1) left it --
    // YT[t][tile][group]: byte offset (walker 0) of the y word of the rank the group produces in the x array of transform
    // t + 1 (by ITS ranks), or in Y for the first transform; Y0T[t]: the same for rank 0.
    {
        auto woff = [](const int r) { return 4 * (((r >> 4) << 8) + ((r & 3) << 6) + ((r >> 2) & 3)); };
        auto src_rank = [&](const int tt, const int g) {
            if (g >= D) return 0;
            return tt == T - 1 ? g : rank_of_feat[(tt + 1) * D + feat_of_rank[tt * D + g]];
        };
        for (int e = threadIdx.x; e < T * (nT + 2) * 4; e += 128) {
            const int tt = e / ((nT + 2) * 4), rem = e - tt * (nT + 2) * 4, tile = rem >> 2, i = rem & 3;
            // (the group's rank, recomputed as the table above does: the table itself is being written by other threads)
            int g = D;
            if (tile < nT) {
                int4 dg = *reinterpret_cast<const int4*>(quad_meta + 4 * tile);
                dg.x &= 0xffff; dg.y &= 0xffff; dg.z &= 0xffff; dg.w &= 0xffff;
                const bool ny = dg.y != dg.x, nz = dg.z != dg.y, nw = dg.w != dg.z;
                const int g1 = ny ? dg.y : (nz ? dg.z : (nw ? dg.w : D));
                const int g2 = ny ? (nz ? dg.z : (nw ? dg.w : D)) : ((nz && nw) ? dg.w : D);
                const int g3 = (ny && nz && nw) ? dg.w : D;
                g = i == 0 ? dg.x : (i == 1 ? g1 : (i == 2 ? g2 : g3));
            }
            YT[e] = woff(src_rank(tt, g));
        }
        for (int tt = threadIdx.x; tt < T; tt += 128) Y0T[tt] = woff(src_rank(tt, 0));
    }
    {   // x arrays, activations (their padding slots are read against zero weights) and staging start zeroed
        float4* z4 = reinterpret_cast<float4*>(XA);
        const int n4 = (2 * Dp * 16 + 3 * Hp * 16 + 2 * NSF2_STAGE_FLOATS + 2 * NSF2_PART_FLOATS) >> 2;
        for (int e = threadIdx.x; e < n4; e += 128) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (wv == 0) {
        if constexpr (FM > 0) {
            for (int e = lane; e < (Dp - D) * 16; e += 64) Y[lidx(D + (e >> 4), e & 15)] = 0.0f;
            const double sg = pa.adapt ? pa.adapt[0] : pa.sigma, ca = pa.adapt ? pa.adapt[1] : pa.cn_a;
            propose_body<FM>(pa.kind, pa.cur32, nullptr, pa.adapt ? pa.adapt + 2 : pa.mu, pa.inv_cov, pa.chol, pa.nu, sg, ca,
                             pa.rng, pa.prop64, nullptr, pa.quad, pa.quad_prop, n, D, Y, rank_of_feat + (T - 1) * D,
                             (int64_t)blockIdx.x);
        } else {
            load_rows(Y, in, row0, n, D, Dp, feat_of_rank + (T - 1) * D, lane);
        }
    }
    float ladj = 0.0f;
    int xsel = 0;
    __syncthreads();

    auto take_table = [&](NsfHid& F, const int tt, const int U) {
        const int4 ty = *reinterpret_cast<const int4*>(YT + (tt * (nT + 2) + U) * 4);
        F.yo[0] = __builtin_amdgcn_readfirstlane(ty.x); F.yo[1] = __builtin_amdgcn_readfirstlane(ty.y);
        F.yo[2] = __builtin_amdgcn_readfirstlane(ty.z); F.yo[3] = __builtin_amdgcn_readfirstlane(ty.w);
        const int4 tg = *reinterpret_cast<const int4*>(DGT + 8 * U);
        const int4 txy = *reinterpret_cast<const int4*>(DGT + 8 * U + 4);
        F.g[0] = __builtin_amdgcn_readfirstlane(tg.x & 0xffff); F.g[1] = __builtin_amdgcn_readfirstlane(tg.y);
        F.g[2] = __builtin_amdgcn_readfirstlane(tg.z); F.g[3] = __builtin_amdgcn_readfirstlane(tg.w);
        F.pat = __builtin_amdgcn_readfirstlane(tg.x >> 16);
        F.xy[0] = __builtin_amdgcn_readfirstlane(txy.x); F.xy[1] = __builtin_amdgcn_readfirstlane(txy.y);
        F.xy[2] = __builtin_amdgcn_readfirstlane(txy.z); F.xy[3] = __builtin_amdgcn_readfirstlane(txy.w);
    };

    if (wv == 1) {
        // ==================================================================================== BURST wave
        // hidden operands of tile TT (two sets, used alternately: tile TT+1's are on their way while TT is prepared)
#define NB_FETCH(TB, TT, P1, P2, XF, Bz0, Bz1, Bz2)                                                               \
        {                                                                                                         \
            const int TT_ = (TT) < nT ? (TT) : nT - 1;                                                            \
            const int so1_ = (TB) + oF1 + TT_ * nT * 1024, so2_ = (TB) + oF2 + TT_ * nT * 1024;                   \
            _Pragma("unroll") for (int i_ = 0; i_ < NSF2_PK; ++i_) {                                              \
                const int vo_ = i_ < TT_ - 1 ? vo_T : NSF2_OOB;                                                   \
                P1[i_] = bload4(rs, vo_, so1_ + i_ * 1024);                                                      \
                P2[i_] = bload4(rs, vo_, so2_ + i_ * 1024);                                                      \
            }                                                                                                     \
            _Pragma("unroll") for (int i_ = 0; i_ < NSF2_PX; ++i_)                                                \
                XF[i_] = bload4(rs, i_ < nXT ? vo_T : NSF2_OOB, (TB) + oF0C + (TT_ * nXT + i_) * 1024);           \
            Bz0 = bload4(rs, vo_q, (TB) + oB0T + 64 * TT_);                                                      \
            Bz1 = bload4(rs, vo_q, (TB) + oB1T + 64 * TT_);                                                      \
            Bz2 = bload4(rs, vo_q, (TB) + oB2T + 64 * TT_);                                                      \
        }
        NsfBurstSet sA, sB;
        const bool static_tiles = nTl <= NSF2_PK + 1;
        NsfBurstCtx bc;
        bc.rs = rs; bc.oF1 = oF1; bc.oF2 = oF2; bc.oF0C = oF0C; bc.oB0T = oB0T; bc.oB1T = oB1T; bc.oB2T = oB2T; bc.oF3I = oF3I; bc.oB3I = oB3I;
        bc.nT = nT; bc.nXT = nXT; bc.D = D; bc.lane = lane; bc.vo_lane = vo_lane; bc.vo_T = vo_T; bc.vo_q = vo_q;
        bc.H0 = H0; bc.H1 = H1; bc.H2 = H2; bc.ts = nullptr;
        // eager partials of the last two live tiles (NSF2_EAGER_OK)
        bc.eag = (NSF2_EAGER_OK(&m) && static_tiles && nTl >= 8 && !(NSF2_ABL & 64) && !(m.reserved & 1)) ? 1 : 0;   // (reserved bit 0: launch_nsf2, PMC_NSF2_EAGER=0)
        const int eT1 = bc.eag ? nTl - 1 : -1, eT2 = bc.eag ? nTl - 2 : -1;
        bc.ea[0] = EAG; bc.ea[1] = EAG + NSF2_PART_FLOATS;
        {
            const int4 t1 = *reinterpret_cast<const int4*>(DGT + 8 * (bc.eag ? nTl - 1 : 0));
            const int4 t2 = *reinterpret_cast<const int4*>(DGT + 8 * (bc.eag ? nTl - 2 : 0));
            bc.eg[0][0] = __builtin_amdgcn_readfirstlane(t1.x & 0xffff); bc.eg[0][1] = __builtin_amdgcn_readfirstlane(t1.y);
            bc.eg[0][2] = __builtin_amdgcn_readfirstlane(t1.z); bc.eg[0][3] = __builtin_amdgcn_readfirstlane(t1.w);
            bc.eg[1][0] = __builtin_amdgcn_readfirstlane(t2.x & 0xffff); bc.eg[1][1] = __builtin_amdgcn_readfirstlane(t2.y);
            bc.eg[1][2] = __builtin_amdgcn_readfirstlane(t2.z); bc.eg[1][3] = __builtin_amdgcn_readfirstlane(t2.w);
        }
        NsfBurstCarry carry;
        if (!static_tiles) {
            NB_FETCH((T - 1) * blk_bytes, 0, sA.p1, sA.p2, sA.xf, sA.b0, sA.b1, sA.b2)
        }
        // (what the x array holds on entry -- zeros, or the x of two transforms ago -- meets zero weights only: the layer-0
        //  fragments f0c carry the columns of the ranks that are final; the other array is the chain's y)
        int spar = 0;                                      // staging parity of the transform's first tile (the buffers alternate across transforms)
        // the static path's first tile of transform tt (biases only): staged while the chain still runs the LAST tile of the
        // transform before, so that at a transform boundary the chain solves rank 0 and goes on
        // the tile after tile U - 1, as the carry sees it: its ranks and the first K step of its own products
        int4 tnn;                                              // the table row `peek` read ahead (its latency passes behind a tile body)
        auto peek = [&](const int U) { tnn = *reinterpret_cast<const int4*>(DGT + 8 * (U < nTl ? U : nT)); };     // (behind the last live tile: "no groups")
        auto set_next = [&](const int U) {                     // (row U was peeked)
            const int4 tn = tnn;
            const bool lv = U < nTl;
            bc.gnx[0] = lv ? (__builtin_amdgcn_readfirstlane(tn.x) & 0xffff) : D; bc.gnx[1] = lv ? __builtin_amdgcn_readfirstlane(tn.y) : D;
            bc.gnx[2] = lv ? __builtin_amdgcn_readfirstlane(tn.z) : D; bc.gnx[3] = lv ? __builtin_amdgcn_readfirstlane(tn.w) : D;
            bc.ksn = (bc.eag && U == eT1) ? 3 : ((bc.eag && U == eT2) ? 2 : 0);
            peek(U + 1);
        };
        auto first_tile = [&](const int tt, float* Xt, const int par) {
            const int4 tg = *reinterpret_cast<const int4*>(DGT);
            bc.tb = tt * blk_bytes; bc.X = Xt;
            bc.g[0] = __builtin_amdgcn_readfirstlane(tg.x & 0xffff); bc.g[1] = __builtin_amdgcn_readfirstlane(tg.y);
            bc.g[2] = __builtin_amdgcn_readfirstlane(tg.z); bc.g[3] = __builtin_amdgcn_readfirstlane(tg.w);
            {   // the first tile's biases (nothing else of it exists): the four ranks' first halves, the two pairs' shared second halves
                const int q_ = lane >> 4;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    carry.b[k] = bload4(rs, bc.g[k] < D ? vo_q : NSF2_OOB, bc.tb + oB3I + (bc.g[k] < D ? bc.g[k] : 0) * 128);
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const int ga = bc.g[2 * pr], gb = bc.g[2 * pr + 1];
                    const int vo = q_ < 2 ? (ga < D ? (q_ << 4) : NSF2_OOB) : (gb < D ? ((q_ - 2) << 4) + (gb - ga) * 128 : NSF2_OOB);
                    carry.b[4 + pr] = bload4(rs, vo, bc.tb + oB3I + (ga < D ? ga : 0) * 128 + 64);
                }
            }
            peek(1);
            set_next(1);
            bc.stg = STG + (par & 1) * NSF2_STAGE_FLOATS;
            bc.part = PART + (par & 1) * NSF2_PART_FLOATS;
            nsf_burst_tile<0>(bc, carry);
        };
        if (static_tiles) first_tile(T - 1, XA, 0);
        for (int t = T - 1; t >= 0; --t) {
            const int tb = t * blk_bytes;
            float* X = xsel ? XB : XA;
            xsel ^= 1;
#define NB_K(NK, P1, P2, AA1, AA2)                                                                                \
            _Pragma("unroll") for (int i_ = 0; i_ < NSF2_PK; ++i_) {                                              \
                if (i_ < (NK)) {                                                                                  \
                    const float4 b1 = *reinterpret_cast<const float4*>(H0 + (i_ << 8) + (lane << 2));             \
                    const float4 b2 = *reinterpret_cast<const float4*>(H1 + (i_ << 8) + (lane << 2));             \
                    AA1 = MFMA(P1[i_].x, b1.x, AA1); AA2 = MFMA(P2[i_].x, b2.x, AA2);                             \
                    AA1 = MFMA(P1[i_].y, b1.y, AA1); AA2 = MFMA(P2[i_].y, b2.y, AA2);                             \
                    AA1 = MFMA(P1[i_].z, b1.z, AA1); AA2 = MFMA(P2[i_].z, b2.z, AA2);                             \
                    AA1 = MFMA(P1[i_].w, b1.w, AA1); AA2 = MFMA(P2[i_].w, b2.w, AA2);                             \
                }                                                                                                 \
            }
            // prepare tile T1 from set (P1 ...) while the chain runs tile T1-1; request tile T1+1 into set (N1 ...)
#define NB_TILE(TT, P1, P2, XF, Bz0, Bz1, Bz2, N1, N2, NXF, Nz0, Nz1, Nz2)                                         \
            {                                                                                                     \
                const int T1 = (TT);                                                                              \
                const int nK = T1 - 1;                          /* hidden tiles 0 .. T1-2 are final */            \
                const int4 tg_ = *reinterpret_cast<const int4*>(DGT + 8 * T1);                                    \
                const int g0_ = __builtin_amdgcn_readfirstlane(tg_.x & 0xffff), g1_ = __builtin_amdgcn_readfirstlane(tg_.y); \
                const int g2_ = __builtin_amdgcn_readfirstlane(tg_.z), g3_ = __builtin_amdgcn_readfirstlane(tg_.w); \
                NB_FETCH(tb, T1 + 1, N1, N2, NXF, Nz0, Nz1, Nz2)                                                  \
                f32x4 a0 = as_acc(Bz0), a1 = as_acc(Bz1), a2 = as_acc(Bz2);                                       \
                if (!(NSF2_ABL & 32)) NB_K(nK, P1, P2, a1, a2)                                                                          \
                nsf_stream2(a1, a2, rs, vo_T, tb + oF1 + T1 * nT * 1024, tb + oF2 + T1 * nT * 1024, NSF2_PK, nK, H0, H1, lane);  /* flows wider than NSF2_PK + 2 tiles */ \
                _Pragma("unroll") for (int i_ = 0; i_ < NSF2_PX; ++i_) {                                          \
                    if (i_ < nXT) {                                                                               \
                        const float4 b = *reinterpret_cast<const float4*>(X + (i_ << 8) + (lane << 2));            \
                        a0 = MFMA(XF[i_].x, b.x, a0); a0 = MFMA(XF[i_].y, b.y, a0);                               \
                        a0 = MFMA(XF[i_].z, b.z, a0); a0 = MFMA(XF[i_].w, b.w, a0);                               \
                    }                                                                                             \
                }                                                                                                 \
                float* st_ = STG + ((T1 + spar) & 1) * NSF2_STAGE_FLOATS;                                         \
                *reinterpret_cast<float4*>(st_ + (lane << 2)) = make_float4(a0[0], a0[1], a0[2], a0[3]);          \
                *reinterpret_cast<float4*>(st_ + 256 + (lane << 2)) = make_float4(a1[0], a1[1], a1[2], a1[3]);    \
                *reinterpret_cast<float4*>(st_ + 512 + (lane << 2)) = make_float4(a2[0], a2[1], a2[2], a2[3]);    \
                float* pt_ = PART + ((T1 + spar) & 1) * NSF2_PART_FLOATS;                                         \
                const int f3_ = tb + oF3I, b3_ = tb + oB3I;                                                       \
                nsf_out_partials_wide(rs, f3_, b3_, nT, D, nK, g0_, H2, pt_, lane, vo_lane, vo_q);                \
                nsf_out_partials_wide(rs, f3_, b3_, nT, D, nK, g1_, H2, pt_ + 512, lane, vo_lane, vo_q);          \
                nsf_out_partials_wide(rs, f3_, b3_, nT, D, nK, g2_, H2, pt_ + 1024, lane, vo_lane, vo_q);         \
                nsf_out_partials_wide(rs, f3_, b3_, nT, D, nK, g3_, H2, pt_ + 1536, lane, vo_lane, vo_q);         \
                lds_bar();                                                /* E(T1 - 1) */                         \
            }
            if (static_tiles) {
                lds_bar_t();                                              // E(-1): the chain solved rank 0
                bc.tb = tb; bc.X = X;
                for (int T1 = 1; T1 < nTl; ++T1) {
                    bc.g[0] = bc.gnx[0]; bc.g[1] = bc.gnx[1]; bc.g[2] = bc.gnx[2]; bc.g[3] = bc.gnx[3];      // (the previous tile's "next")
                    set_next(T1 + 1);
                    bc.stg = STG + ((T1 + spar) & 1) * NSF2_STAGE_FLOATS;
                    bc.part = PART + ((T1 + spar) & 1) * NSF2_PART_FLOATS;
#ifdef NSF2_TILE_STAMPS
                    bc.ts = ts ? ts + 4 * T * (nTl + 1) + 4 * bar_no : nullptr;       // (behind the barrier stamps: four section stamps per tile)
#endif
                    switch (T1) {
#define CASE(K) case K: nsf_burst_tile<K>(bc, carry); break;
#define CASE_E(K) case K: if (T1 == eT1) nsf_burst_tile<K, 3>(bc, carry); else if (T1 == eT2) nsf_burst_tile<K, 2>(bc, carry); \
                          else nsf_burst_tile<K>(bc, carry); break;
#define CASE_J(K) case K: if (bc.eag) nsf_burst_tile<K, 0, true>(bc, carry); else nsf_burst_tile<K>(bc, carry); break;
                        CASE(1) CASE_J(2) CASE_J(3) CASE_J(4) CASE(5) CASE_E(6) CASE_E(7) CASE_E(8) CASE_E(9) CASE_E(10)
#undef CASE_J
#undef CASE_E
#undef CASE
                        default: break;
                    }
                    lds_bar_t();                                          // E(T1 - 1)
                }
                if (t > 0) first_tile(t - 1, xsel ? XB : XA, spar + nTl);     // (the next transform's x array and parity)
            } else {
                for (int T2 = 0; T2 < nTl; T2 += 2) {
                    NB_TILE(T2, sA.p1, sA.p2, sA.xf, sA.b0, sA.b1, sA.b2, sB.p1, sB.p2, sB.xf, sB.b0, sB.b1, sB.b2)
                    if (T2 + 1 >= nTl) break;
                    NB_TILE(T2 + 1, sB.p1, sB.p2, sB.xf, sB.b0, sB.b1, sB.b2, sA.p1, sA.p2, sA.xf, sA.b0, sA.b1, sA.b2)
                }
                const int tbn = (t > 0 ? t - 1 : 0) * blk_bytes;      // the next transform's first operands: on their way before this one ends
                NB_FETCH(tbn, 0, sA.p1, sA.p2, sA.xf, sA.b0, sA.b1, sA.b2)
            }
            if (static_tiles) lds_bar_t(); else lds_bar();                // E(nTl - 1)
            spar = (spar + nTl) & 1;
            if (t == 0) __syncthreads();                                  // (the chain stored the result)
        }
#undef NB_TILE
#undef NB_K
#undef NB_FETCH
    } else {
        // ==================================================================================== CHAIN wave
        NsfHid fA, fB;
        NsfOut ob[2];
        // the chain's hidden operands of tile U of transform tt
        // request number K of 11 of the chain's hidden operands of tile U of transform tt
        auto request_hid = [&](NsfHid& F, auto k_, const int tt, const int U) {
            constexpr int K = decltype(k_)::value;
            const int base = tt * blk_bytes;
            const int Un = U + 1 < nT ? U + 1 : U;
            const int voN = U + 1 < nT ? vo_T : NSF2_OOB;
            if constexpr (K == 0) F.wt1 = bload4(rs, vo_T, base + oF1 + (U * nT + U) * 1024);
            else if constexpr (K == 1) F.wt2 = bload4(rs, vo_T, base + oF2 + (U * nT + U) * 1024);
            else if constexpr (K == 2) F.wn1 = bload4(rs, voN, base + oF1 + (Un * nT + U) * 1024);
            else if constexpr (K == 3) F.wn2 = bload4(rs, voN, base + oF2 + (Un * nT + U) * 1024);
            else if constexpr (K < 7) F.w0o[K - 4] = bload4(rs, ((4 + K - 4) << 6) + vo_q, base + oCW0 + U * 1024);
            else if constexpr (K < 11) F.w0N[K - 7] = bload4(rs, U + 1 < nT ? ((K - 7) << 6) + vo_q : NSF2_OOB, base + oCW0 + Un * 1024);
        };
        // request number K of 4 of rank g's two output tiles against hidden tiles Kp (previous; < 0: none) and Kc (own) of transform tt
        auto request_out = [&](NsfOut& O, auto k_, const int tt, const int g, const int Kp, const int Kc) {
            constexpr int K = decltype(k_)::value;
            const bool lv = g < D;
            const int so = tt * blk_bytes + oF3I + (lv ? g : 0) * 2 * nT * 1024;
            const int vp = (lv && Kp >= 0) ? vo_lane : NSF2_OOB, vc = lv ? vo_lane : NSF2_OOB;
            const int kp = Kp >= 0 ? Kp : 0;
            if (NSF2_ABL & 16) return;
            if constexpr (K == 0) O.fp0 = bload4(rs, vp, so + kp * 1024);
            else if constexpr (K == 1) O.fp1 = bload4(rs, vp, so + (nT + kp) * 1024);
            else if constexpr (K == 2) O.fc0 = bload4(rs, vc, so + Kc * 1024);
            else O.fc1 = bload4(rs, vc, so + (nT + Kc) * 1024);
        };
        take_table(fA, T - 1, 0);
        fA.w0o[3] = fB.w0o[3] = make_float4(0.f, 0.f, 0.f, 0.f);          // (the fourth group has no later quad)
        static_for<11>([&](auto k_) { request_hid(fA, k_, T - 1, 0); });
        static_for<4>([&](auto k_) { request_out(ob[0], k_, T - 1, fA.g[0], -1, 0); });
        // the group after the first: the first tile's second group, or (a tile of one group) the second tile's first
        auto after_first = [&](const int tt, const int U, const int patU, const int g1U, int& tt2, int& U2, int& g2) {
            const bool two = (patU & (patU - 1)) != 0;
            const bool more2 = U + 1 < nTl;
            const int stt = more2 ? tt : (tt > 0 ? tt - 1 : 0), sU = more2 ? U + 1 : 0;
            tt2 = two ? tt : stt;
            U2 = two ? U : sU;
            g2 = two ? g1U : (__builtin_amdgcn_readfirstlane(DGT[8 * sU]) & 0xffff);
        };
        {
            int tt2, U2, g2;
            after_first(T - 1, 0, fA.pat, fA.g[1], tt2, U2, g2);
            static_for<4>([&](auto k_) { request_out(ob[1], k_, tt2, g2, U2 - 1, U2); });
        }
        NsfChain s;
        s.o0 = s.o1 = f32x4{0.f, 0.f, 0.f, 0.f};                          // (the first tile of the sweep has no previous tile)
        float4 w00 = bload4(rs, vo_q, (T - 1) * blk_bytes + oCW0);      // layer 0, first tile: the column of rank 0
        float4 r00 = bload4(rs, vo_q, (T - 1) * blk_bytes + oB3I), r01 = bload4(rs, vo_q, (T - 1) * blk_bytes + oB3I + 64);
        const float* Ysrc = Y;                             // the input of the transform: Y, then the previous transform's x array
        int spar = 0;
        for (int t = T - 1; t >= 0; --t) {
            float* X = xsel ? XB : XA;
            xsel ^= 1;
            {   // rank 0 reads nothing: bias only
                float xv, l;
                const float y0 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(Ysrc) + (p << 4) + Y0T[t]);
                rqs_inverse_split(as_acc(r00), as_acc(r01), PAR + (p << 5), q, y0, xv, l);
                if (q == 0) X[lidx(0, p)] = xv;
                ladj -= l;
                WAVE_LDS_FENCE();
                s.a0N[0] = w00.x * xv; s.a0N[1] = w00.y * xv; s.a0N[2] = w00.z * xv; s.a0N[3] = w00.w * xv;
            }
            s.accN1 = s.accN2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) s.h2s[j] = 0.0f;
            lds_bar_t();                                                  // E(-1): the first tile's staging is complete

            for (int Tt_ = 0; Tt_ < nTl; ++Tt_) {
                const int Tt = __builtin_amdgcn_readfirstlane(Tt_);
                if (pf) pf[15] = clock64();
                NsfHid& cur = fA;
                NsfHid& nxt = fB;
                const float* st = STG + ((Tt + spar) & 1) * NSF2_STAGE_FLOATS;
                const float* part = PART + ((Tt + spar) & 1) * NSF2_PART_FLOATS;
                const float4 s0 = *reinterpret_cast<const float4*>(st + (lane << 2));
                const float4 s1 = *reinterpret_cast<const float4*>(st + 256 + (lane << 2));
                const float4 s2 = *reinterpret_cast<const float4*>(st + 512 + (lane << 2));
                s.a0[0] = s0.x + s.a0N[0]; s.a0[1] = s0.y + s.a0N[1]; s.a0[2] = s0.z + s.a0N[2]; s.a0[3] = s0.w + s.a0N[3];
                s.p1[0] = s1.x + s.accN1[0]; s.p1[1] = s1.y + s.accN1[1]; s.p1[2] = s1.z + s.accN1[2]; s.p1[3] = s1.w + s.accN1[3];
                s.p2[0] = s2.x + s.accN2[0]; s.p2[1] = s2.y + s.accN2[1]; s.p2[2] = s2.z + s.accN2[2]; s.p2[3] = s2.w + s.accN2[3];
#pragma unroll
                for (int j = 0; j < 4; ++j) { s.a0N[j] = 0.0f; s.h2p[j] = s.h2s[j]; s.h0s[j] = s.h1s[j] = s.h2s[j] = 0.0f; }
                s.accN1 = s.accN2 = s.acc1 = s.acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
                {   // the first group's parameters: what the previous tile's last group formed in the shadow of its spline solve
                    // (this tile's "previous tile" products) + the staged partial (bias + h2 tiles <= Tt-2)
                    const float4 q0 = *reinterpret_cast<const float4*>(part + (lane << 2));
                    const float4 q1 = *reinterpret_cast<const float4*>(part + 256 + (lane << 2));
                    s.o0[0] += q0.x; s.o0[1] += q0.y; s.o0[2] += q0.z; s.o0[3] += q0.w;
                    s.o1[0] += q1.x; s.o1[1] += q1.y; s.o1[2] += q1.z; s.o1[3] += q1.w;
                }
                const int pat = cur.pat;
                // what follows this tile: the next live tile, or the first tile of the next transform
                const bool more = Tt + 1 < nTl;
                const int ntt = more ? t : (t > 0 ? t - 1 : 0), nU = more ? Tt + 1 : 0;
                take_table(nxt, ntt, nU);
                int tt2, U2, g2;                             // the group after the next tile's first
                after_first(ntt, nU, nxt.pat, nxt.g[1], tt2, U2, g2);
                // the shadows of a group's MFMAs (slot 0..3: behind the pairs of previous-tile output products; 4, 5: behind the
                // hidden hops): the next group's output fragments -- a later group of this tile, or the next tile's first -- and
                // this group's share of the next tile's hidden operands
                auto ahead = [&](auto gi_, auto slot_, auto ng_) {
                    constexpr int G = decltype(gi_)::value, SL = decltype(slot_)::value, NG_ = decltype(ng_)::value;
                    if constexpr (SL < 4 && G == NG_) {
                        request_out(ob[1], slot_, tt2, g2, U2 - 1, U2);
                    } else if constexpr (SL < 4) {
                        const bool last = G + 1 >= NG_;
                        const int gn = last ? nxt.g[0] : cur.g[(G + 1) & 3];
                        request_out(ob[(G + 1) & 1], slot_, last ? ntt : t, gn, last ? (more ? Tt : -1) : Tt - 1, last ? nU : Tt);
                    } else {
                        constexpr int LPG = (11 + NG_ - 1) / NG_, H = (LPG + 1) / 2;       // loads per group; in the first hop's shadow
                        constexpr int k0 = G * LPG + (SL == 4 ? 0 : H), k1 = G * LPG + (SL == 4 ? H : LPG);
                        static_for<k1 - k0>([&](auto j_) {
                            constexpr int K = k0 + decltype(j_)::value;
                            if constexpr (K < 11) request_hid(nxt, std::integral_constant<int, K>{}, ntt, nU);
                        });
                    }
                };
                switch (pat) {
#define CASE(P) case P: nsf_group<P, 0>(s, cur, ob, part, X, Ysrc, PAR, D, q, p, lane, ladj, ahead, pf); break;
                    CASE(1) CASE(3) CASE(5) CASE(7) CASE(9) CASE(11) CASE(13) CASE(15)
#undef CASE
                    default: break;
                }
                NSF_STAMP(6)
                {   // the tile's activations: one 16-byte word per layer and lane (row q of every quad)
                    const int hw = (Tt << 8) + (q << 6) + (p << 2);
                    *reinterpret_cast<float4*>(H0 + hw) = make_float4(s.h0s[0], s.h0s[1], s.h0s[2], s.h0s[3]);
                    *reinterpret_cast<float4*>(H1 + hw) = make_float4(s.h1s[0], s.h1s[1], s.h1s[2], s.h1s[3]);
                    *reinterpret_cast<float4*>(H2 + hw) = make_float4(s.h2s[0], s.h2s[1], s.h2s[2], s.h2s[3]);
                }
                NSF_STAMP(8)
                lds_bar_t();                                 // E(Tt): this tile is final
                NSF_STAMP(9)
                fA = fB;
                NSF_STAMP(10)
            }
            w00 = bload4(rs, vo_q, (t > 0 ? t - 1 : 0) * blk_bytes + oCW0);
            r00 = bload4(rs, vo_q, (t > 0 ? t - 1 : 0) * blk_bytes + oB3I);
            r01 = bload4(rs, vo_q, (t > 0 ? t - 1 : 0) * blk_bytes + oB3I + 64);
            Ysrc = X;                                      // the next transform reads its y from here, through its offset table
            spar = (spar + nTl) & 1;
            if (t == 0) {
                for (int e = lane; e < D * 16; e += 64) {
                    const int r = e >> 4, pp = e & 15;
                    if (row0 + pp < n) out[(row0 + pp) * D + PRM[r]] = X[lidx(r, pp)];
                }
                __syncthreads();
            }
        }
        if (ladj_out && lane < 16 && row0 + p < n) ladj_out[row0 + p] = ladj;
        if (pf && lane == 0) for (int i = 0; i < 16; ++i) prof[i] = pfv[i];
    }
    if constexpr (FM > 0) {
        // the scaler (+ prior) on the 16 walkers while they are in LDS (x of the first transform, by its ranks); both
        // wavefronts; the activation arrays are free now and serve as its scratch
        if (pa.epi.on) {
            const float* Xl = (xsel ^ 1) ? XB : XA;
            scaler_epilogue(pa.epi, Xl, rank_of_feat, reinterpret_cast<double*>(H0), row0, n, D, (int)threadIdx.x, 128,
                            [](int r, int pp) { return lidx(r, pp); });
        }
    }
}

// Covered: spline flows whose degree groups fit a hidden tile (tri_ok), D <= 64 (NSF2_PX x tiles), one buffer resource
// over the whole image.  -1: not covered (the caller launches the lone-wave sweep / the separate kernels).
// pa == nullptr: plain inverse of z; else the fused proposal (+ scaler epilogue when pa->epi.on).
static int launch_nsf2(const ProposeArgs* pa, const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n, hipStream_t stream) {
    static const int mode = pmc_env_int("PMC_INVERSE_NSF_DUO", -1);
    if (mode == 0) return -1;
    if (m->n_out != 23 || !m->tri_ok || m->D > 64 || m->D < 2) return -1;
    if (m->pk_per_transform * 4 * m->T >= (int64_t)NSF2_OOB) return -1;
    const size_t lds = (size_t)NSF2_LDS_FLOATS(m) * sizeof(float);
    if (lds > 160 * 1024) return -1;
    const ProposeArgs none{};
    // PMC_MAF_VARIANT_LEFT_LOOKING: the burst wave's left-looking schedule without the eager partials (the two must agree
    // bit for bit: every partial receives its K tiles in ascending order either way; tests/test_gpu_flow.py)
    pmc_maf_t mk = *m;
    mk.reserved = (m->reserved & PMC_MAF_VARIANT_LEFT_LOOKING) ? 1 : 0;
#define LAUNCHN(FMV)                                                                                              \
    {                                                                                                             \
        if (lds > 48 * 1024) {                                                                                    \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(maf_inverse_nsf2_kernel<FMV>),       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
            if (e != hipSuccess) return pmc_fail_hip(e, "hipFuncSetAttribute(maf_inverse_nsf2_kernel)");          \
        }                                                                                                         \
        hipLaunchKernelGGL(maf_inverse_nsf2_kernel<FMV>, dim3((unsigned)((n + 15) / 16)), dim3(128), lds, stream, mk, z, x, \
                           ladj, n, pa ? pa->prof : (long long*)nullptr, pa ? *pa : none);                        \
    }
    if (pa && pa->prof) LAUNCHN(-1)
    else if (!pa || !pa->cur32) LAUNCHN(0)
    else if (m->D <= 16) LAUNCHN(4)
    else if (m->D <= 32) LAUNCHN(8)
    else LAUNCHN(16)
#undef LAUNCHN
    return pmc_check_launch("maf_inverse_nsf2_kernel");
}

// whether PMC_INVERSE_AUTO launches this kernel for the flow (bench.py names the kernel it times with it)
extern "C" int pmc_maf_inverse_auto_is_nsf2(const pmc_maf_t* m) {
    static const int mode = pmc_env_int("PMC_INVERSE_NSF_DUO", -1);
    if (!m || mode == 0 || m->n_out != 23 || !m->tri_ok || m->D > 64 || m->D < 2) return 0;
    if (m->pk_per_transform * 4 * m->T >= (int64_t)NSF2_OOB) return 0;
    return (size_t)NSF2_LDS_FLOATS(m) * sizeof(float) <= 160 * 1024 ? 1 : 0;
}

int pmc_launch_inverse_nsf2(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n, hipStream_t stream) {
    return launch_nsf2(nullptr, m, z, x, ladj, n, stream);
}

// the spline flows' instance of pmc_launch_propose_inverse_tri4 (same contract; called from there)
int pmc_launch_propose_inverse_nsf2(ProposeArgs* pa, const ScalerEpi* epi, int* epi_done, const pmc_maf_t* m, float* x, float* ladj,
                                    int64_t n, hipStream_t stream) {
    if (epi_done) *epi_done = 0;
    // the scaler as the sweep's epilogue: its scratch aliases the three activation arrays of the walker set
    const bool epi_ok = epi && epi_done && epi->s.D == m->D && scaler_epilogue_lds_bytes(m->D) <= (size_t)3 * m->Hp * 16 * sizeof(float);
    if (epi_ok) { pa->epi = *epi; pa->epi.on = 1; }
    const int rc = launch_nsf2(pa, m, nullptr, x, ladj, n, stream);
    if (rc == 0 && epi_ok) *epi_done = 1;
    return rc;
}

#ifdef PMC_DEBUG_HOOKS
// measurement only (scripts/profile_nsf2.py): cycles of the chain wave of workgroup 0 by section of a group, summed over the sweep
extern "C" int pmc_debug_nsf2_profile(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n, long long* prof, void* stream) {
    ProposeArgs pa{};
    pa.prof = prof;
    return launch_nsf2(&pa, m, z, x, ladj, n, (hipStream_t)stream) < 0 ? pmc_fail("pmc_debug_nsf2_profile: flow not covered") : 0;
}
#endif

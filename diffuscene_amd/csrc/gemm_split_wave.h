// Wave-autonomous form of the split-bf16 GEMM (round 6): the same arithmetic, the same accumulator layout and -- per output element --
// the same sequence of MFMAs as gemm_split.hip (results are bit-identical), but NO block-level staging and NO block barrier:
//
//   * a wave owns one scene (<= 16 RB token rows) x 128 channels = RB x 8 MFMA blocks of 16 x 16 (two GroupNorm cells), one wave per
//     SIMD (the 512-register budget: 160 accumulators + two register sets of weight fragments);
//   * the weight planes arrive FRAGMENT-MAJOR (dsc_split_bf16x3_f32 with layout = DSC_PLANES_FRAGMENT: [3][n/16][K/32][64 lanes][8]):
//     one buffer_load_dwordx4 per (16-channel block, plane, K tile) brings a wave's MFMA fragment straight from L2 into registers,
//     1 KiB contiguous, one K tile ahead in the second register set -- no LDS, no DMA issue stalls, no fragment reads;
//   * the token rows are loaded by the wave itself (coalesced: 4 lanes x 32 B per row), split exactly in registers under the MFMAs of
//     the previous tile and turned into MFMA fragments through a WAVE-PRIVATE LDS image (two stages x 3 planes x 16 RB rows x 64 B):
//     a wave's LDS instructions execute in order, so nothing but the wave's own counters orders a write before the read that needs it.
//
// What this removes from the block-staged kernel: 16 block barriers per launch with their top-of-tile bubbles (fragments can only be
// read after the barrier there), the LDS-DMA issue stalls, 12 of 27 fragment reads per K tile, and the lock-step of eight waves that
// reach their epilogues in the same cycle.  What it costs: every scene's rows are split by the four waves that own its four channel
// quarters (0.92 VALU per MFMA instead of 0.37), and the weight fragments cross the L2 -> CU path once per wave.
#pragma once
#include "dsc_common.h"
#include <utility>

// s_memtime stamp hooks of tools/wave_probe.hip (no-ops in the product build)
#ifndef DSC_WAVE_STAMP_ARGS
#define DSC_WAVE_STAMP_ARGS
#define DSC_WAVE_STAMP_PASS
#define DSC_WAVE_STAMP_DECL
#define DSC_WAVE_STAMP_KERNEL(id)
#define DSC_WAVE_STAMP_TILE(id)
#define DSC_WAVE_STAMP_FLUSH(kt)
#define DSC_WAVE_STAMP_END
#endif

namespace dsc_wave {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 32;
constexpr int WAVES = 4;          // per block: four scenes x one 128-channel chunk (the waves share weight lines in the CU's L1)

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// exact 3-way split of 8 consecutive f32 into three packed bf16x8 fragments (identical to dsc_split::split8)
__device__ __forceinline__ void split8(const f32x4 lo, const f32x4 hi, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
    u32x4 a, b, c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x0 = q < 2 ? lo[2 * q] : hi[2 * q - 4], x1 = q < 2 ? lo[2 * q + 1] : hi[2 * q - 3];
        const unsigned u1 = cvt_pk_bf16(x0, x1);
        const float r0 = x0 - bf_lo(u1), r1 = x1 - bf_hi(u1);
        const unsigned u2 = cvt_pk_bf16(r0, r1);
        const float s0 = r0 - bf_lo(u2), s1 = r1 - bf_hi(u2);
        a[q] = u1;
        b[q] = u2;
        c[q] = cvt_pk_bf16(s0, s1);
    }
    p1 = __builtin_bit_cast(bf16x8, a);
    p2 = __builtin_bit_cast(bf16x8, b);
    p3 = __builtin_bit_cast(bf16x8, c);
}

template <int RB>
constexpr int wave_smem_bytes() { return WAVES * 2 * 3 * (16 * RB) * 64; }

// GN: the fused Block epilogue (GroupNorm + (scale, shift) + SiLU); PERROW: its (scale, shift) rows are per token / per slot
// J: 16-channel MFMA blocks per wave -- 8 (128 channels, two GroupNorm cells: the launches that fill the chip with 4 waves per CU) or 4 (64
// channels, one cell: half-size launches, e.g. B = 128 scenes of 80 tokens x 512 channels = 1024 waves of 80 x 64)
// GB (dense form only): GroupNorm-backward epilogue -- the product is the gradient w.r.t. a fused Block's output, the kernel writes the gradient w.r.t.
// its pre-norm activation (dsc_gemm_args.gnb_*)
template <bool GN, int RB, bool PERROW = false, int J = 8, bool GB = false>
__global__ __launch_bounds__(64 * WAVES, 1) void gemm_split_wave_kernel(const dsc_gemm_args p, const uint16_t* const planes_f, const int ntok DSC_WAVE_STAMP_ARGS) {
    static_assert(J == 8 || J == 4, "wave tile: 128 or 64 channels");
    static_assert(!GB || (!GN && J == 8), "the GroupNorm-backward epilogue rides on the dense 80 x 128 form");
    constexpr int NWF = 3 * J;                            // weight fragments per K tile and wave
    constexpr int CW = 16 * J;                            // channels per wave
    constexpr int ROWS = 16 * RB, XP = ROWS * 64, STAGE = 3 * XP;
    constexpr int NWB = RB > 4 ? RB - 2 : RB > 3 ? RB - 1 : RB;   // token blocks that carry weight loads: the last ones are spared when there are enough
                                                                  // (a fragment is needed at the top of the next tile: >= 2 blocks of lead)
    constexpr int PERW = (NWF + NWB - 1) / NWB;
    __shared__ __attribute__((aligned(16))) char smem[wave_smem_bytes<RB>()];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K = p.k1 + p.k2, KT = K / BK;
    const int scenes = (p.m + ntok - 1) / ntok;
    const int sgs = (scenes + WAVES - 1) / WAVES, chunks = p.n / CW;
    // XCD-aware order (block b runs on XCD b % 8): the channel chunks of a scene group sit on one XCD (its L2 serves the token rows
    // to the three other chunks)
    const int bid = blockIdx.x;
    int sg, ch;
    if ((sgs & 7) == 0) {
        const int xcd = bid & 7, idx = bid >> 3;
        sg = xcd * (sgs >> 3) + idx / chunks;
        ch = idx % chunks;
    } else {
        sg = bid / chunks;
        ch = bid % chunks;
    }
    const int scene = sg * WAVES + wave;
    if (scene >= scenes) return;                          // no block barrier anywhere: a wave without a scene just leaves
    DSC_WAVE_STAMP_DECL
    DSC_WAVE_STAMP_KERNEL(0)
    const int z = blockIdx.y;
    const int row0 = scene * ntok, col0 = ch * CW;
    const int rows_here = min(ntok, p.m - row0);
    const float* const xb1 = p.a1 + (int64_t)z * p.sa1 + (int64_t)row0 * p.lda1;
    const float* const xb2 = p.a2 ? p.a2 + (int64_t)z * p.sa2 + (int64_t)row0 * p.lda2 : xb1;
    const float* const bias = p.bias ? p.bias + (int64_t)z * p.sbias : nullptr;
    // descriptors end with the operand: rows past the last one (the padding rows of a ragged last scene) read zeros
    const int64_t rec1 = ((int64_t)(p.m - row0 - 1) * p.lda1 + p.k1) * 4;
    const int64_t rec2 = p.a2 ? ((int64_t)(p.m - row0 - 1) * p.lda2 + p.k2) * 4 : rec1;
    const int64_t recw = (int64_t)3 * p.batch * p.n * K * 2;
    const int64_t rec1c = rec1 < 0x7fffffffLL ? rec1 : 0x7fffffffLL, rec2c = rec2 < 0x7fffffffLL ? rec2 : 0x7fffffffLL;
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb1), 0, (int)rec1c, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb2), 0, (int)rec2c, 0x00020000);
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(planes_f), 0, (int)recw, 0x00020000);
#endif
    const int plane_bytes = p.batch * p.n * K * 2;
    const int nb0 = (z * p.n + col0) / 16;               // first 16-channel block of this wave in the stacked planes

    // per-lane offsets: token staging item = (row lane >> 2 of the block, k-octet lane & 3); fragment = (row lane & 15, k-octet lane >> 4)
    const int q4 = lane & 3, r16 = lane >> 2, g = lane >> 4, l15 = lane & 15;
    const int tvo1 = (r16 * (int)p.lda1 + q4 * 8) * 4, tvo2 = (r16 * (int)p.lda2 + q4 * 8) * 4;
    const int ildso = r16 * 64 + ((q4 ^ ((r16 >> 1) & 3)) << 4);
    const int xoff = l15 * 64 + ((g ^ ((l15 >> 1) & 3)) << 4);
    const int wvo = lane * 16;
    char* const wl = smem + wave * (2 * STAGE);

    f32x4 acc[RB][J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (bias) b4 = *reinterpret_cast<const f32x4*>(bias + col0 + j * 16 + 4 * g);
#pragma unroll
        for (int i = 0; i < RB; ++i) acc[i][j] = b4;
    }

    f32x4 ld[RB][2];
    bf16x8 wf[2][J][3], xf[2][3];
    // Token rows of tile kt, block u, from K segment SEG (compile time: a run-time segment select inside the K loop costs a branch or a
    // descriptor rebuilt through memory).  Segment 1 of a single-segment launch is segment 0 again.
    const int lda2e = p.a2 ? (int)p.lda2 : (int)p.lda1, kshift = p.a2 ? p.k1 : 0;
    const int tvo2e = p.a2 ? tvo2 : tvo1;
    auto load_tok = [&](int kt, int u, auto segc) {
#if defined(__HIP_DEVICE_COMPILE__)
        constexpr int SEG = decltype(segc)::value;
        const int k0 = kt * BK;
        if constexpr (SEG == 0) {
            const int so = (u * 16 * (int)p.lda1 + k0) * 4;
            ld[u][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1, tvo1, so, 0));
            ld[u][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1, tvo1 + 16, so, 0));
        } else {
            const int so = (u * 16 * lda2e + k0 - kshift) * 4;
            ld[u][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2, tvo2e, so, 0));
            ld[u][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2, tvo2e + 16, so, 0));
        }
#else
        (void)kt; (void)u;
#endif
    };
    auto load_w = [&](int kt, int j, int pl) -> bf16x8 {
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wvo, pl * plane_bytes + ((nb0 + j) * KT + kt) * 1024, 0));
#else
        (void)kt; (void)j; (void)pl;
        return bf16x8{};
#endif
    };
    constexpr int NMMA = 6 * J;

    // One K tile.  Registers on entry: wf[CUR] = weight fragments of tile kt, ld = f32 token rows of tile kt + 1; LDS stage CUR = token
    // fragments of tile kt.  During the tile: wf[NXT] <- tile kt + 1, stage NXT <- split(ld), ld <- tile kt + 2.
    //
    // The instruction stream of a token block is placed BY HAND: every MFMA is followed by its "filler" (at most two VALU, or one LDS
    // / global access) and a full scheduling barrier, so the program order below IS the issue order.  (Left to the scheduler -- with or
    // without sched_group_barrier recipes -- the 48 MFMAs of a block come out accumulator-major, i.e. dependent MFMAs back to back,
    // and the split clumps into one VALU run.)  Filler slots of block i, after MFMA number k:
    //   k = 0..2                 the three fragment reads of block i + 1
    //   k = 3..3 + PERW - 1      this block's share of the weight loads of tile kt + 1
    //   then 22 slots            the split of ld[i] (44 VALU, two per slot, two element pairs interleaved)
    //   then 3 slots             its three LDS plane writes
    //   then 1 slot              the two token loads of tile kt + 2 that refill ld[i]
    auto tile = [&](const int kt, auto curc, auto segc) {
        constexpr int CUR = decltype(curc)::value, NXT = CUR ^ 1;
        char* const cur = wl + CUR * STAGE;
        char* const nxt = wl + NXT * STAGE;
        const int kn1 = min(kt + 1, KT - 1), kn2 = min(kt + 2, KT - 1);      // the tail re-loads the last tile (no branch in the loop)
        // token fragments: block i of this tile sits in xf[(X0 + i) & 1]; the last block reads block 0 of the NEXT tile (stage NXT, written
        // during this tile's block 0) into the other set, so a tile starts with its first fragments in registers
        constexpr int X0 = CUR ? (RB & 1) : 0;
        DSC_WAVE_STAMP_TILE(0)
        auto block = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int NR = 3;
            constexpr int W0 = i * PERW < NWF ? i * PERW : NWF, W1 = (i + 1) * PERW < NWF ? (i + 1) * PERW : NWF;
            constexpr int VPS = J == 8 ? 2 : 4, NVS = 44 / VPS;               // VALU of the split per filler slot (24 MFMAs per block at J = 4)
            constexpr int S0 = NR + (W1 - W0), S1 = S0 + NVS, S2 = S1 + 3, S3 = S2 + 1;
            static_assert(S3 <= NMMA, "one MFMA per filler slot");
            // split state: pair q = elements 2q, 2q + 1 of the lane's 8; op o of the 11-op chain (see split8)
            unsigned su1[4], su2[4], su3[4];
            float sr0[4], sr1[4], st0[4], st1[4];
            auto xin = [&](int q, int h) -> float { return q < 2 ? ld[i][0][2 * q + h] : ld[i][1][2 * q - 4 + h]; };
            auto split_op = [&](auto nc) {
                constexpr int n = decltype(nc)::value;                      // 0..43: pairs (0,1) interleaved, then (2,3)
                constexpr int q = 2 * (n / 22) + (n & 1), o = (n % 22) >> 1;
                if constexpr (o == 0) su1[q] = cvt_pk_bf16(xin(q, 0), xin(q, 1));
                if constexpr (o == 1) st0[q] = bf_lo(su1[q]);
                if constexpr (o == 2) st1[q] = bf_hi(su1[q]);
                if constexpr (o == 3) sr0[q] = xin(q, 0) - st0[q];
                if constexpr (o == 4) sr1[q] = xin(q, 1) - st1[q];
                if constexpr (o == 5) su2[q] = cvt_pk_bf16(sr0[q], sr1[q]);
                if constexpr (o == 6) st0[q] = bf_lo(su2[q]);
                if constexpr (o == 7) st1[q] = bf_hi(su2[q]);
                if constexpr (o == 8) sr0[q] = sr0[q] - st0[q];
                if constexpr (o == 9) sr1[q] = sr1[q] - st1[q];
                if constexpr (o == 10) su3[q] = cvt_pk_bf16(sr0[q], sr1[q]);
            };
            char* const d0 = nxt + i * 1024 + ildso;
            auto filler = [&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if constexpr (k < NR) {
                    if constexpr (i + 1 < RB) xf[(X0 + i + 1) & 1][k] = *reinterpret_cast<const bf16x8*>(cur + k * XP + (i + 1) * 1024 + xoff);
                    else xf[(X0 + i + 1) & 1][k] = *reinterpret_cast<const bf16x8*>(nxt + k * XP + xoff);
                } else if constexpr (k < S0) {
                    constexpr int w = W0 + (k - NR);
                    wf[NXT][w / 3][w % 3] = load_w(kn1, w / 3, w % 3);
                } else if constexpr (k < S1) {
                    dsc_static_for<VPS>([&](auto vc) { split_op(std::integral_constant<int, VPS * (k - S0) + decltype(vc)::value>{}); });
                } else if constexpr (k < S2) {
                    constexpr int pl = k - S1;
                    const u32x4 v = pl == 0 ? u32x4{su1[0], su1[1], su1[2], su1[3]} : pl == 1 ? u32x4{su2[0], su2[1], su2[2], su2[3]}
                                                                                              : u32x4{su3[0], su3[1], su3[2], su3[3]};
                    *reinterpret_cast<u32x4*>(d0 + pl * XP) = v;
                } else if constexpr (k == S2) {
                    load_tok(kn2, i, segc);
                }
            };
            // product-major, small terms first: per accumulator the SAME sequence as dsc_split::gemm_split_tile
            auto product = [&](auto pc) {
                constexpr int pr = decltype(pc)::value;
                constexpr int wp = pr == 0 ? 2 : (pr == 1 || pr >= 4) ? 0 : 1;          // w3, w1, w2, w2, w1, w1
                constexpr int xp = pr == 0 ? 0 : pr == 1 ? 2 : pr == 2 ? 1 : pr == 3 ? 0 : pr == 4 ? 1 : 0;   // x1, x3, x2, x1, x2, x1
                dsc_static_for<J>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[CUR][j][wp], xf[(X0 + i) & 1][xp], acc[i][j], 0, 0, 0);
                    filler(std::integral_constant<int, pr * J + j>{});
                    __builtin_amdgcn_sched_barrier(0);
                });
            };
            dsc_static_for<6>(product);
            DSC_WAVE_STAMP_TILE(1 + i)
        };
        dsc_static_for<RB>(block);
        DSC_WAVE_STAMP_FLUSH(kt)
    };

    // prologue: weights of tile 0, token rows of tile 0 -> stage 0, token rows of tile 1 (both in segment 0: k1 >= 64)
    using std::integral_constant;
#pragma unroll
    for (int u = 0; u < RB; ++u) load_tok(0, u, integral_constant<int, 0>{});
#pragma unroll
    for (int w = 0; w < NWF; ++w) wf[0][w / 3][w % 3] = load_w(0, w / 3, w % 3);
    // Every CU's waves are in this prologue at once and the memory system serves the burst at ~11 B per cycle and CU: what is requested
    // before the first MFMA IS the prologue.  So the rows of tile 1 are requested block by block BEHIND the split of tile 0 (8.4 k cycles;
    // requesting them up front, "one round trip instead of two", measured 10 k: profiles/r06_wave_probe_4.txt vs _5.txt)
#pragma unroll
    for (int u = 0; u < RB; ++u) {
        bf16x8 a, b, c;
        split8(ld[u][0], ld[u][1], a, b, c);
        char* d0 = wl + u * 1024 + ildso;
        *reinterpret_cast<bf16x8*>(d0) = a;
        *reinterpret_cast<bf16x8*>(d0 + XP) = b;
        *reinterpret_cast<bf16x8*>(d0 + 2 * XP) = c;
        load_tok(1, u, integral_constant<int, 0>{});
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) xf[0][pl] = *reinterpret_cast<const bf16x8*>(wl + pl * XP + xoff);
    // KT and k1 / 32 are even (the dispatcher requires k1 % 64 == 0 and k2 % 64 == 0): tiles whose look-ahead (kt + 2) still lies in
    // segment 0, then the rest (look-ahead in segment 1, clamped to the last tile)
    const int KT1 = p.k1 / BK;
    int kt = 0;
    DSC_WAVE_STAMP_KERNEL(1)
    for (; kt + 2 < KT1; kt += 2) {
        tile(kt, integral_constant<int, 0>{}, integral_constant<int, 0>{});
        tile(kt + 1, integral_constant<int, 1>{}, integral_constant<int, 0>{});
    }
    // GroupNorm form: the per-channel parameters of the epilogue's row layout (this lane's four channels of each of the two cells: gamma,
    // beta, per-scene (scale, shift)) are requested HERE, before the last tiles of the K loop -- scattered 16-byte loads under full memory
    // load take 1-2 k cycles, and the epilogue's statistics are too short to cover them (32 registers, free at this point)
    f32x4 ep_ga[2], ep_be[2], ep_sc[2], ep_sh[2];
    if constexpr (GN) {
        const bool per_scene_e = p.ss_mode == DSC_SS_PER_SCENE || p.ss_mode == DSC_SS_BY_INDEX;
        int64_t ssrow_e = scene;
        if (p.ss_mode == DSC_SS_BY_INDEX) ssrow_e = dsc_clamp_index(p.ss_index[scene], p.ss_rows);
#pragma unroll
        for (int c2 = 0; c2 < J / 4; ++c2) {
            const int ch = col0 + 64 * c2 + 4 * (lane & 15);
            ep_ga[c2] = *reinterpret_cast<const f32x4*>(p.gamma + ch);
            ep_be[c2] = *reinterpret_cast<const f32x4*>(p.beta + ch);
            ep_sc[c2] = f32x4{0.f, 0.f, 0.f, 0.f};
            ep_sh[c2] = ep_sc[c2];
            if (per_scene_e) {
                const float* ssr = p.scale_shift + ssrow_e * p.ld_ss + ch;
                ep_sc[c2] = *reinterpret_cast<const f32x4*>(ssr);
                ep_sh[c2] = *reinterpret_cast<const f32x4*>(ssr + p.n);
            }
        }
    }
    for (; kt < KT; kt += 2) {
        tile(kt, integral_constant<int, 0>{}, integral_constant<int, 1>{});
        tile(kt + 1, integral_constant<int, 1>{}, integral_constant<int, 1>{});
    }

    DSC_WAVE_STAMP_KERNEL(2)
    // Epilogue.  Accumulator layout: lane = token (block i, row l15) x 4 consecutive channels (4 g ..) per MFMA block j -- a store from it
    // touches 64-byte pieces of 16 different rows (measured: ~5 B per cycle and CU).  So results leave through the wave's now idle LDS
    // image, 64 channels (one GroupNorm cell) at a time: the raw accumulators are written in the accumulator layout and read back in a
    // ROW layout (lane = row 4 q + er, 16-byte column ec of the cell's 256 B), in which every global access of the epilogue -- result,
    // residual, saved pre-activation, per-row (scale, shift), activation-gradient operand -- moves 4 rows x 256 contiguous bytes, and
    // all element-wise arithmetic runs on per-lane channel constants.  Per element the operations and their order are those of
    // gemm_split.hip: bit-identical results.  Rows past the operand (padding rows of the last block) fall outside the buffer
    // descriptors: their loads return zeros, their stores are dropped.
    constexpr int ERS = 64 * 4 + 16;                      // row stride of the cell image (16 B pad: conflict-free column writes)
    static_assert(ROWS * ERS <= 2 * STAGE, "the transposed cell fits the wave's LDS image");
    const int er = lane >> 4, ec = lane & 15;
#if defined(__HIP_DEVICE_COMPILE__)
    auto row_rsrc = [&](const float* base, int64_t ld, int rows, int cols) {   // base = (row 0, column 0) of this wave's rows
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, rows > 0 ? (int)(((int64_t)(rows - 1) * ld + cols) * 4) : 0, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t ry = row_rsrc(p.y + (int64_t)z * p.sy + (int64_t)row0 * p.ldy, p.ldy, rows_here, p.n);
    const __amdgpu_buffer_rsrc_t rr = row_rsrc(p.residual ? p.residual + (int64_t)z * p.sres + (int64_t)row0 * p.ldr : p.y, p.ldr, p.residual ? rows_here : 0, p.n);
    const __amdgpu_buffer_rsrc_t rpa = row_rsrc(p.preact ? p.preact + (int64_t)z * p.sy + (int64_t)row0 * p.ld_preact : p.y, p.ld_preact, p.preact ? rows_here : 0, p.n);
    auto ld4 = [&](const __amdgpu_buffer_rsrc_t& r, int voff, int soff) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); };
    auto st4 = [&](const f32x4 v, const __amdgpu_buffer_rsrc_t& r, int voff, int soff) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0); };
#else
    const int ry = 0, rr = 0, rpa = 0;
    auto ld4 = [&](int, int, int) { return f32x4{}; };
    auto st4 = [&](const f32x4, int, int, int) {};
#endif
    const bool has_res = p.residual != nullptr, has_pre = p.preact != nullptr;
    auto to_lds = [&](const int c2) {                     // raw accumulators of cell c2 (a compile-time constant at every call), accumulator layout
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(wl + (i * 16 + l15) * ERS + (j * 16 + 4 * g) * 4) = acc[i][4 * c2 + j];
    };
    // Row-layout passes work on QH row groups (4 rows each) at a time: all their global loads first (residual, per-row (scale, shift),
    // activation-gradient operand: one latency per half cell instead of one per row group), then LDS reads, arithmetic, stores.
    constexpr int NQ = ROWS / 4, QH = NQ % 10 == 0 ? 10 : NQ % 8 == 0 ? 8 : NQ % 6 == 0 ? 6 : 4;
    static_assert(NQ % QH == 0, "row groups per pass");
    if constexpr (GB) {
        // GroupNorm-backward epilogue (the arithmetic of train.hip: gn_silu_bwd_reg_kernel, per cell of one scene x 64 channels, here in the row layout: a lane holds
        // rows 4 q + er, four channels): dh = the product (from the LDS image), z = the Block's saved pre-norm activation (whole 256-byte row pieces).
        //   mu, rs      statistics of z over the cell (two wave sums)
        //   u = (ga xh + be) (1 + scale) + shift,  du = dh silu'(u),  dgh = du (1 + scale),  dxh = dgh ga
        //   dz = rs (dxh - mean(dxh) - xh mean(dxh xh))                                      (two more wave sums)
        //   per-scene column sums: dgamma += dgh xh, dbeta += dgh, dbias += dz, d scale += du gh, d shift += du
        const float inv_cnt = 1.f / (float)(ntok * 64);
        const bool per_scene = p.ss_mode == DSC_SS_PER_SCENE && p.scale_shift;
#if defined(__HIP_DEVICE_COMPILE__)
        const __amdgpu_buffer_rsrc_t rz = row_rsrc(p.gnb_z + (int64_t)row0 * p.ld_gnb_z, p.ld_gnb_z, rows_here, p.n);
#else
        const int rz = 0;
#endif
        dsc_static_for<2>([&](auto c2c) {
            constexpr int c2 = decltype(c2c)::value;
            const int ch = col0 + 64 * c2 + 4 * ec, cc = ch * 4;
            const int vy = er * (int)p.ldy * 4 + cc, vz = er * (int)p.ld_gnb_z * 4 + cc;
            f32x4 zv[NQ], dv[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) zv[q] = ld4(rz, vz, 4 * q * (int)p.ld_gnb_z * 4);
            const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + ch), be = *reinterpret_cast<const f32x4*>(p.beta + ch);
            f32x4 s1 = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (per_scene) {
                const float* ssr = p.scale_shift + (int64_t)scene * p.ld_ss + ch;
                s1 = *reinterpret_cast<const f32x4*>(ssr) + 1.f;
                sh = *reinterpret_cast<const f32x4*>(ssr + p.n);
            }
            to_lds(c2);
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                dv[q] = *reinterpret_cast<const f32x4*>(wl + (4 * q + er) * ERS + ec * 16);
                const bool ok = 4 * q + er < rows_here;              // (rows past the scene: z read zeros, the product is padding -- neither may count)
                if (!ok) dv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                s += ok ? (zv[q][0] + zv[q][1]) + (zv[q][2] + zv[q][3]) : 0.f;
            }
            const float mu = wave_sum_dpp(s) * inv_cnt;
            s = 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const bool ok = 4 * q + er < rows_here;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = ok ? zv[q][e] - mu : 0.f; s = fmaf(d, d, s); }
            }
            const float rs = 1.0f / sqrtf(wave_sum_dpp(s) * inv_cnt + p.eps);
            float S1 = 0.f, S2 = 0.f;
            f32x4 Gg = {0.f, 0.f, 0.f, 0.f}, Gb = Gg, Gsc = Gg, Gsh = Gg, Gz = Gg;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (zv[q][e] - mu) * rs;
                    const float gh = ga[e] * xh + be[e];
                    const float u = gh * s1[e] + sh[e];
                    const float sig = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
                    const float du = dv[q][e] * (sig * (1.0f + u * (1.0f - sig)));
                    const float dgh = du * s1[e];
                    const float dxh = dgh * ga[e];
                    S1 += dxh; S2 += dxh * xh;
                    Gg[e] += dgh * xh; Gb[e] += dgh; Gsc[e] += du * gh; Gsh[e] += du;
                    zv[q][e] = xh;
                    dv[q][e] = dxh;
                }
            }
            const float m1 = wave_sum_dpp(S1) * inv_cnt, m2 = wave_sum_dpp(S2) * inv_cnt;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const bool ok = 4 * q + er < rows_here;
                f32x4 dzv;
#pragma unroll
                for (int e = 0; e < 4; ++e) { dzv[e] = rs * (dv[q][e] - m1 - zv[q][e] * m2); Gz[e] += ok ? dzv[e] : 0.f; }
                st4(dzv, ry, vy, 4 * q * (int)p.ldy * 4);
            }
            // column sums over the scene's rows: in-lane over q (above), then over the four row lanes er that share a column quad
            auto fold = [&](f32x4 v) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += __shfl_xor(v[e], 16, 64); v[e] += __shfl_xor(v[e], 32, 64); }
                return v;
            };
            Gg = fold(Gg); Gb = fold(Gb); Gz = fold(Gz);
            if (per_scene && p.gnb_dss) { Gsc = fold(Gsc); Gsh = fold(Gsh); }
            if (lane < 16) {
                const int64_t o = (int64_t)scene * p.gnb_pstride + ch;
                *reinterpret_cast<f32x4*>(p.gnb_dgamma + o) = Gg;
                *reinterpret_cast<f32x4*>(p.gnb_dbeta + o) = Gb;
                *reinterpret_cast<f32x4*>(p.gnb_dbias + o) = Gz;
                if (per_scene && p.gnb_dss) {
                    *reinterpret_cast<f32x4*>(p.gnb_dss + (int64_t)scene * p.ld_gnb_dss + ch) = Gsc;
                    *reinterpret_cast<f32x4*>(p.gnb_dss + (int64_t)scene * p.ld_gnb_dss + p.n + ch) = Gsh;
                }
            }
        });
    } else if constexpr (!GN) {
#if defined(__HIP_DEVICE_COMPILE__)
        const __amdgpu_buffer_rsrc_t rg = row_rsrc(p.actgrad_x ? p.actgrad_x + (int64_t)z * p.sy + (int64_t)row0 * p.ld_actgrad : p.y, p.ld_actgrad, p.actgrad_x ? rows_here : 0, p.n);
#else
        const int rg = 0;
#endif
        // MODE 0: y = u; 1: y = act(u); 2: y = u * act'(actgrad_x); ACT: the activation (compile time inside the loops)
        auto cell = [&](auto c2c, auto modec, auto actc) {     // (c2 at compile time: a run-time index into the accumulators would send them to scratch)
            constexpr int c2 = decltype(c2c)::value, MODE = decltype(modec)::value, ACT = decltype(actc)::value;
            const int cc = (col0 + 64 * c2 + 4 * ec) * 4;
            const int vy = er * (int)p.ldy * 4 + cc, vr = er * (int)p.ldr * 4 + cc, vp = er * (int)p.ld_preact * 4 + cc, vg = er * (int)p.ld_actgrad * 4 + cc;
            to_lds(c2);
#pragma unroll 1
            for (int h = 0; h < NQ; h += QH) {
                f32x4 rres[QH], gx[QH];
                if (has_res) {
#pragma unroll
                    for (int qq = 0; qq < QH; ++qq) rres[qq] = ld4(rr, vr, 4 * (h + qq) * (int)p.ldr * 4);
                }
                if constexpr (MODE == 2) {
#pragma unroll
                    for (int qq = 0; qq < QH; ++qq) gx[qq] = ld4(rg, vg, 4 * (h + qq) * (int)p.ld_actgrad * 4);
                }
#pragma unroll
                for (int qq = 0; qq < QH; ++qq) {
                    f32x4 y = *reinterpret_cast<const f32x4*>(wl + (4 * (h + qq) + er) * ERS + ec * 16);
                    if (has_pre) st4(y, rpa, vp, 4 * (h + qq) * (int)p.ld_preact * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (MODE == 1) y[e] = dsc_act(y[e], ACT);
                        if constexpr (MODE == 2) y[e] *= dsc_act_grad(gx[qq][e], ACT);
                    }
                    if (has_res) y += rres[qq];
                    st4(y, ry, vy, 4 * (h + qq) * (int)p.ldy * 4);
                }
            }
        };
        auto cells = [&](auto modec, auto actc) {
            cell(std::integral_constant<int, 0>{}, modec, actc);
            if constexpr (J == 8) cell(std::integral_constant<int, 1>{}, modec, actc);
        };
        using std::integral_constant;
        const int act = p.act_out;
        if (p.actgrad_x) {
            if (act == DSC_ACT_GELU) cells(integral_constant<int, 2>{}, integral_constant<int, DSC_ACT_GELU>{});
            else if (act == DSC_ACT_SILU) cells(integral_constant<int, 2>{}, integral_constant<int, DSC_ACT_SILU>{});
            else if (act == DSC_ACT_LEAKY01) cells(integral_constant<int, 2>{}, integral_constant<int, DSC_ACT_LEAKY01>{});
            else cells(integral_constant<int, 0>{}, integral_constant<int, DSC_ACT_NONE>{});      // act'(x) = 1
        } else {
            if (act == DSC_ACT_GELU) cells(integral_constant<int, 1>{}, integral_constant<int, DSC_ACT_GELU>{});
            else if (act == DSC_ACT_SILU) cells(integral_constant<int, 1>{}, integral_constant<int, DSC_ACT_SILU>{});
            else if (act == DSC_ACT_LEAKY01) cells(integral_constant<int, 1>{}, integral_constant<int, DSC_ACT_LEAKY01>{});
            else cells(integral_constant<int, 0>{}, integral_constant<int, DSC_ACT_NONE>{});
        }
    } else {
#if defined(__HIP_DEVICE_COMPILE__)
        // per-row (scale, shift): row = token (PER_TOKEN) or slot (PER_SLOT), [scale(n) | shift(n)] per row
        const __amdgpu_buffer_rsrc_t rs = row_rsrc(PERROW ? p.scale_shift + (p.ss_mode == DSC_SS_PER_TOKEN ? (int64_t)row0 * p.ld_ss : 0) : p.y, p.ld_ss,
                                                   PERROW ? rows_here : 0, 2 * p.n);
#else
        const int rs = 0;
#endif
        const float inv_cnt = 1.f / (float)(ntok * 64);
        // GroupNorm rows are whole scenes (m % ntok == 0): every block but the last is full
        const bool vlast = (RB - 1) * 16 + l15 < rows_here;
#pragma unroll
        for (int c2 = 0; c2 < J / 4; ++c2) {              // the GroupNorm cells of the wave (two at J = 8): channels col0 + 64 c2 .. + 63
            const int ch = col0 + 64 * c2 + 4 * ec;       // this lane's four channels in the row layout
            const f32x4 ga = ep_ga[c2], be = ep_be[c2], sc = ep_sc[c2], sh = ep_sh[c2];      // (requested before the last K tiles)
            to_lds(c2);
            float s0 = 0.f;
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 a = acc[i][4 * c2 + j];
                    const float t = (a[0] + a[1]) + (a[2] + a[3]);
                    s0 += (i < RB - 1 || vlast) ? t : 0.f;
                }
            const float mean = wave_sum_dpp(s0) * inv_cnt;
            float q0 = 0.f;
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d = (i < RB - 1 || vlast) ? acc[i][4 * c2 + j][e] - mean : 0.f;
                        q0 = fmaf(d, d, q0);
                    }
            const float rstd = 1.f / sqrtf(wave_sum_dpp(q0) * inv_cnt + p.eps);
            if (c2 == 0) { DSC_WAVE_STAMP_KERNEL(3) }
            f32x4 A, B;                                   // y = z * A + B with A = rstd*gamma*(scale+1), B = (beta - mean*rstd*gamma)*(scale+1) + shift
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = rstd * ga[e], sp1 = sc[e] + 1.f;
                A[e] = a * sp1;
                B[e] = fmaf(be[e] - mean * a, sp1, sh[e]);
            }
            const int cc = ch * 4;
            const int vy = er * (int)p.ldy * 4 + cc, vr = er * (int)p.ldr * 4 + cc, vp = er * (int)p.ld_preact * 4 + cc, vs = er * (int)p.ld_ss * 4 + cc;
            if (c2 == 0) { DSC_WAVE_STAMP_KERNEL(4) }
#pragma unroll
            for (int h = 0; h < NQ; h += QH) {
                f32x4 rres[QH], s1[QH], s2[QH];
                if (has_res) {
#pragma unroll
                    for (int qq = 0; qq < QH; ++qq) rres[qq] = ld4(rr, vr, 4 * (h + qq) * (int)p.ldr * 4);
                }
                if constexpr (PERROW) {
#pragma unroll
                    for (int qq = 0; qq < QH; ++qq) {
                        s1[qq] = ld4(rs, vs, 4 * (h + qq) * (int)p.ld_ss * 4);
                        s2[qq] = ld4(rs, vs + p.n * 4, 4 * (h + qq) * (int)p.ld_ss * 4);
                    }
                }
#pragma unroll
                for (int qq = 0; qq < QH; ++qq) {
                    const f32x4 zv = *reinterpret_cast<const f32x4*>(wl + (4 * (h + qq) + er) * ERS + ec * 16);
                    if (has_pre) st4(zv, rpa, vp, 4 * (h + qq) * (int)p.ld_preact * 4);
                    f32x4 y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = fmaf(zv[e], A[e], B[e]);
                        if constexpr (PERROW) t = fmaf(t, s1[qq][e] + 1.f, s2[qq][e]);
                        y[e] = dsc_silu_fast(t);
                    }
                    if (has_res) y += rres[qq];
                    st4(y, ry, vy, 4 * (h + qq) * (int)p.ldy * 4);
                }
            }
            if (c2 == 0) { DSC_WAVE_STAMP_KERNEL(5) }
        }
    }
    DSC_WAVE_STAMP_END
}

template <bool GN, int RB, bool PERROW = false, int J = 8, bool GB = false>
int launch(const dsc_gemm_args* a, const uint16_t* planes_f, int ntok, hipStream_t s) {
    const int scenes = (a->m + ntok - 1) / ntok;
    const unsigned grid = (unsigned)(((scenes + WAVES - 1) / WAVES) * (a->n / (16 * J)));
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL((gemm_split_wave_kernel<GN, RB, PERROW, J, GB>), dim3(grid, (unsigned)a->batch), dim3(64 * WAVES), 0, s, *a, planes_f, ntok DSC_WAVE_STAMP_PASS);
    DSC_LAUNCH_CHECK();
    return 0;
}

}  // namespace dsc_wave

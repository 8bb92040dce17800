// f32-accurate GEMM on the bf16 matrix cores ("split-bf16", 3 pieces x 6 products), the fast path behind dsc_gemm_f32 /
// dsc_gemm_gn_silu_f32 when the caller supplies pre-split weight planes (dsc_gemm_args.w_planes).
//
//   y[m][n] = epilogue( bias[n] + sum_k [a1 | a2][m][k] * w[n][k] )            a1, a2, w, y: f32 in HBM, exactly as the f32-MFMA path
//
// Arithmetic.  Every f32 operand is split EXACTLY into three bf16 pieces, x = x1 + x2 + x3 (round-to-nearest at each step: 8 + 8 + 8
// mantissa bits = the 24 of an f32), every bf16 x bf16 product is exact in f32, and the six products whose weight is >= 2^-18 of the
// leading one are accumulated in f32 by v_mfma_f32_16x16x32_bf16:
//
//   x*w ~= x1*w1 + x1*w2 + x2*w1 + x2*w2 + x1*w3 + x3*w1            (dropped: x2*w3, x3*w2, x3*w3 <= 2^-24 relative)
//
// Measured on MI355X against an f64 product (profiles/r03_bf16x6_*.txt): max / rms error <= those of the exact-f32 MFMA kernel at
// K = 128 .. 3072 (the MFMA sums 32 products per instruction before it rounds into the accumulator: 16 roundings per K = 512
// instead of 256).  The bf16 pipe is 16x the f32-MFMA rate, six products are 2.67x the f32-MFMA roofline; measured, the launch runs at
// ~1.5x the f32-MFMA kernel (79.5 -> 52.7 us at M = 20480, n = K = 512) with the matrix cores busy 0.43 of the time AT FULL CLOCK
// (GRBM_GUI_ACTIVE / 8 over the duration = 2.36 GHz, profiles/r05_gemm_gn_hbm_traffic.json): the K loop of this block-staged form is
// issue- and barrier-bound (77 % issue efficiency inside the loop, ~20 % exposed prologue + store burst), not power-bound -- only the
// grouped weight-gradient launch is throttled (1.86 GHz at busy 0.68).  Round 6 adds the wave-autonomous family (gemm_split_wave.h):
// same arithmetic bit for bit, no block barrier, row-layout epilogue; it takes the launches that come out at whole waves per SIMD.
//
// Layout (the best of the forms measured in round 3, "PIPE 3" of the experiment): 8 waves as WM x WN; a wave owns ONE scene (<= 16*RB
// tokens, padded to RB MFMA row blocks inside LDS) x 64 channels = RB x 4 MFMA blocks of 16 x 16 -- exactly one GroupNorm cell, so the
// fused Block epilogue needs no LDS and no block barrier.  The weights arrive pre-split as three bf16 planes [3][n][k]
// (dsc_split_bf16x3_f32, once per weight update) and are staged by LDS-DMA (lane-linear image, XOR swizzle applied on the global
// side).  The tokens are split ONCE per block: the staging threads load the f32 tile of K tile kt+1 into registers, split it under
// the MFMAs of tile kt and write three bf16 planes to the other LDS stage.  The MFMA computes out^T (weights as the row operand) so
// that a lane holds 4 consecutive channels of one token: 16-byte stores.
#include "dsc_common.h"
#include "gemm_split_wave.h"
#include <atomic>
#include <cstring>

#ifndef DSC_SPLIT_DSPREAD
#define DSC_SPLIT_DSPREAD true
#endif

namespace dsc_split {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 32;

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));     // v_cvt_pk_bf16_f32, round to nearest even
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// exact 3-way split of 8 consecutive f32 into three packed bf16x8 fragments
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

// ---------------------------------------------------------------------------------------------------------------------
// weight planes: out[p][r][c] (p = 0..2, [rows_out][cols_out] bf16 each) from w [rows][cols] f32, optionally transposed
// ---------------------------------------------------------------------------------------------------------------------
struct SplitBatch { dsc_split_item it[DSC_WS_MAX]; };

__global__ __launch_bounds__(256) void split_planes_kernel(const SplitBatch bch) {
    dsc_split_item it = bch.it[blockIdx.y];
    const bool frag = (it.transpose & DSC_SPLIT_FRAGMENT) != 0;                                    // fragment-major output (gemm_split_wave.h)
    it.transpose &= DSC_SPLIT_TRANSPOSE;
    const int ro = it.transpose ? it.cols : it.rows, co = it.transpose ? it.rows : it.cols;       // output matrix
    const int oct = co / 8;                                                                        // 8-element items per output row
    const long items = (long)ro * oct;
    const long plane = (long)ro * co;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (long)gridDim.x * blockDim.x) {
        // consecutive threads take consecutive output rows of one octet when transposing (their reads are then consecutive
        // input columns), consecutive octets of one row otherwise
        const long r = it.transpose ? i % ro : i / oct;
        const long o = it.transpose ? i / ro : i % oct;
        f32x4 lo, hi;
        if (it.transpose) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lo[e] = it.w[(o * 8 + e) * it.ldw + r];
                hi[e] = it.w[(o * 8 + 4 + e) * it.ldw + r];
            }
        } else {
            lo = *reinterpret_cast<const f32x4*>(it.w + r * it.ldw + o * 8);
            hi = *reinterpret_cast<const f32x4*>(it.w + r * it.ldw + o * 8 + 4);
        }
        bf16x8 a, b, c;
        split8(lo, hi, a, b, c);
        // fragment-major: [ro / 16][co / 32][lane = (k-octet & 3) * 16 + row % 16][8]
        uint16_t* d = it.planes + (frag ? ((((r >> 4) * (co >> 5) + (o >> 2)) * 64 + (o & 3) * 16 + (r & 15)) * 8) : (r * co + o * 8));
        *reinterpret_cast<bf16x8*>(d) = a;
        *reinterpret_cast<bf16x8*>(d + plane) = b;
        *reinterpret_cast<bf16x8*>(d + 2 * plane) = c;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS bytes of a WM x WN x RB tile: two stages of (3 token planes + 3 weight planes) + the dump slot of a ragged item round
template <int WM, int WN, int RB>
constexpr int split_smem_bytes() {
    constexpr int BM = 16 * RB * WM, BN = 64 * WN;
    constexpr int STAGE = 3 * BM * BK * 2 + 3 * BN * BK * 2;
    constexpr int ITEMS_W = BM * 4 / (WM * WN);
    return 2 * STAGE + ((ITEMS_W % 64) ? 1024 : 0);
}

// One output tile of one launch: block `bid` of problem `zb` (the body of gemm_split_kernel; a device function so that a persistent
// launch can run several tiles / layers per block -- tools/two_layer_probe.hip, the round-4 GO / NO-GO experiment).
struct NoSync { __device__ __forceinline__ void operator()() const {} };

// `sync` runs after the weight DMA of the first K tile has been issued and before the first token rows are read: a persistent launch
// waits there for the producer of its input rows (the weights do not depend on it).
template <bool GN, int WM, int WN, int RB, bool DSPREAD, class Sync = NoSync>
__device__ __forceinline__ void gemm_split_tile(const dsc_gemm_args& p, const int ntok, const int bid, const int zb, char* const smem,
                                                const Sync sync = Sync{}) {
    constexpr int NW = WM * WN, T = 64 * NW;
    static_assert(NW == 8 || NW == 4, "4 or 8 waves");
    constexpr int BM = 16 * RB * WM, BN = 64 * WN;       // BM = LDS rows (scenes padded to 16*RB); global rows = WM * ntok
    constexpr int B_PLANE = BN * BK * 2, X_PLANE = BM * BK * 2;
    constexpr int XA = 3 * X_PLANE, STAGE = XA + 3 * B_PLANE;
    constexpr int CH_PL = B_PLANE / 1024, CH = 3 * CH_PL, NI = (CH + NW - 1) / NW;      // one-KiB wave transfers of the weight planes
    constexpr int ITEMS_W = BM * 4 / NW;                 // (row, k-octet) staging items per wave
    constexpr int NIT = (ITEMS_W + 63) / 64;             // per lane
    constexpr int DUMP = (ITEMS_W % 64) ? 1024 : 0;      // where the idle lanes of a ragged item round write
    static_assert(NIT <= RB && NIT <= 3, "the splits ride in the last NIT token blocks of a tile");
    static_assert(2 * STAGE + DUMP <= 160 * 1024, "two stages must fit the 160 KiB LDS");
    static_assert(2 * STAGE + DUMP == split_smem_bytes<WM, WN, RB>(), "split_smem_bytes");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave_u % WM, wn = wave_u / WM;
    const int K = p.k1 + p.k2;
    // XCD-aware block order: the channel blocks of a token block sit next to each other on one XCD (shared token rows in its L2)
    const int scenes = (p.m + ntok - 1) / ntok;
    const int cbs = p.n / BN, rbs = (scenes + WM - 1) / WM;
    int rb, cb;
    if ((rbs & 7) == 0) {
        const int xcd = bid & 7, idx = bid >> 3;
        rb = xcd * (rbs >> 3) + idx / cbs;
        cb = idx % cbs;
    } else {
        rb = bid / cbs;
        cb = bid % cbs;
    }
    const int row0 = rb * WM * ntok, col0 = cb * BN;     // first global token row of the block
    const int rows_here = p.m - row0;                    // valid global rows from row0 on (may exceed the block)
    // grouped launch (batch > 1): problem z = zb (zb); its weights are rows [z n, (z+1) n) of ONE stacked matrix whose planes
    // are [3][batch n][K]
    const int z = zb;
    const float* const xb1 = p.a1 + (int64_t)z * p.sa1 + (int64_t)row0 * p.lda1;
    const float* const xb2 = p.a2 ? p.a2 + (int64_t)z * p.sa2 + (int64_t)row0 * p.lda2 : xb1;
    const uint16_t* const wb = p.w_planes + ((int64_t)z * p.n + col0) * K;
    const int plane_bytes = p.batch * p.n * K * 2;
    const float* const bias = p.bias ? p.bias + (int64_t)z * p.sbias : nullptr;

    // weight-plane DMA: chunk c = 16 channel rows x 64 B of one plane; k-octet g lands in slot g ^ ((n >> 1) & 3)
    int dvoff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int c = wave_u + NW * i;
        if (c >= CH) c -= CH;                            // a ragged last round re-fetches chunks 0..: same bytes
        const int plane = c / CH_PL, nrow = (c % CH_PL) * 16 + (lane >> 2);
        dvoff[i] = plane * plane_bytes + nrow * K * 2 + (((lane & 3) ^ ((nrow >> 1) & 3)) << 4);
    }
    // chunks [i0, i1) of this wave's share of one weight tile
    auto dma_part = [&](int kt, char* stage, auto i0c, auto i1c) {
        constexpr int i0 = decltype(i0c)::value, i1 = decltype(i1c)::value;
        __attribute__((address_space(3))) char* lbase = (__attribute__((address_space(3))) char*)stage;
#pragma unroll
        for (int i = i0; i < (i1 < NI ? i1 : NI); ++i) {
            int c = wave_u + NW * i;
            if (c >= CH) c -= CH;
#if defined(__HIP_DEVICE_COMPILE__)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)const_cast<uint16_t*>(wb), 0, 0x7fffffff, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lbase + XA + c * 1024, 16, dvoff[i], kt * BK * 2, 0, 0);
#else
            (void)lbase; (void)kt;
#endif
        }
    };
    auto dma_tile = [&](int kt, char* stage) { dma_part(kt, stage, std::integral_constant<int, 0>{}, std::integral_constant<int, NI>{}); };
    // DSPREAD: the weight DMA of the next tile is issued PER chunks per token block, between the block's MFMAs, instead of all NI
    // chunks back to back at the top of the tile (an LDS-DMA instruction holds a wave's issue for 60-185 cycles, and the two waves
    // of a SIMD reach the top of the tile together)
    constexpr int PER = DSPREAD ? (NI + RB - 1) / RB : NI;

    // accumulators: out^T blocks, lane = (token lane&15, channels 4*(lane>>4) .. +3); bias folded into the initial value
    f32x4 acc[RB][4];
    const int g = lane >> 4, l15 = lane & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (bias) b4 = *reinterpret_cast<const f32x4*>(bias + col0 + wn * 64 + j * 16 + 4 * g);
#pragma unroll
        for (int i = 0; i < RB; ++i) acc[i][j] = b4;
    }

    int woff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nr = wn * 64 + j * 16 + l15;
        woff[j] = XA + nr * 64 + ((g ^ ((nr >> 1) & 3)) << 4);
    }
    auto mma_block = [&](const bf16x8 (&wf)[4][3], const bf16x8& x1, const bf16x8& x2, const bf16x8& x3, f32x4 (&c)[4]) {
        // product-major: the same accumulator comes round every 4th MFMA; small terms first
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][2], x1, c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], x3, c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][1], x2, c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][1], x1, c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], x2, c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], x1, c[j], 0, 0, 0);
    };
    constexpr int NMMA = 24;                             // MFMAs per 16-token block

    // token staging: item = (LDS row r, k-octet q): 8 consecutive k of one token row; LDS row r = scene r / (16 RB), token
    // r % (16 RB); padding rows re-read row 0 (never stored)
    int ioff1[NIT], ioff2[NIT], ildso[NIT];
    bool ion[NIT];
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int idx = 64 * u + lane;
        ion[u] = idx < ITEMS_W;
        const int t = wave_u * ITEMS_W + (ion[u] ? idx : 0);
        const int r = t >> 2, q = t & 3;
        const int sc = r / (16 * RB), tk = r % (16 * RB), gr = sc * ntok + tk;
        const int grow = (tk < ntok && gr < rows_here) ? gr : 0;
        ioff1[u] = grow * (int)p.lda1 * 4 + q * 32;
        ioff2[u] = grow * (int)p.lda2 * 4 + q * 32;
        ildso[u] = r * 64 + ((q ^ ((r >> 1) & 3)) << 4);
    }
    int xoff[RB];                                        // token fragment of block i: row (wm*RB+i)*16 + l15, k-octet g
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int r = (wm * RB + i) * 16 + l15;
        xoff[i] = r * 64 + ((g ^ ((r >> 1) & 3)) << 4);
    }
    f32x4 ld[NIT][2];
    auto load_items = [&](int kt) {
        const int k0 = kt * BK;
        const bool seg1 = k0 < p.k1;
        const char* base = reinterpret_cast<const char*>(seg1 ? xb1 : xb2) + (seg1 ? k0 : k0 - p.k1) * 4;
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int o = seg1 ? ioff1[u] : ioff2[u];
            ld[u][0] = *reinterpret_cast<const f32x4*>(base + o);
            ld[u][1] = *reinterpret_cast<const f32x4*>(base + o + 16);
        }
    };
    auto store_item = [&](int u, char* stage) {
        bf16x8 a, b, c;
        split8(ld[u][0], ld[u][1], a, b, c);
        // idle lanes of a ragged round (they hold a copy of item 0) write to a dump slot instead of being masked off: an
        // exec-masked store would cut the MFMA stream into basic blocks
        char* d0 = ion[u] ? stage + ildso[u] : smem + 2 * STAGE + lane * 16;
        const int ps = ion[u] ? X_PLANE : 0;
        *reinterpret_cast<bf16x8*>(d0) = a;
        *reinterpret_cast<bf16x8*>(d0 + ps) = b;
        *reinterpret_cast<bf16x8*>(d0 + 2 * ps) = c;
    };

    const int KT = K / BK;
    dma_tile(0, smem);
    sync();
    load_items(0);
#pragma unroll
    for (int u = 0; u < NIT; ++u) store_item(u, smem);
    for (int kt = 0; kt < KT; ++kt) {
        __builtin_amdgcn_s_waitcnt(0x0070);              // vmcnt(0) lgkmcnt(0): my DMA chunks and my plane writes of tile kt are done
        __syncthreads();                                 // everyone's are; nobody reads the other stage any more
        char* cur = smem + (kt & 1) * STAGE;
        char* nxt = smem + ((kt + 1) & 1) * STAGE;
        const int kn = min(kt + 1, KT - 1);              // the tail re-stages the last tile (no branch in the loop)
        if constexpr (!DSPREAD) dma_tile(kn, nxt);
        load_items(kn);
        bf16x8 wf[4][3], xf[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) xf[0][pl] = *reinterpret_cast<const bf16x8*>(cur + pl * X_PLANE + xoff[0]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wf[j][pl] = *reinterpret_cast<const bf16x8*>(cur + woff[j] + pl * B_PLANE);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (i + 1 < RB) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) xf[(i + 1) & 1][pl] = *reinterpret_cast<const bf16x8*>(cur + pl * X_PLANE + xoff[i + 1]);
            }
            if (i == RB - NIT) {
                // the staged f32 rows of the next tile have arrived; with DSPREAD they are the OLDEST requests of the tile (the
                // weight DMA chunks were issued after them and may still fly: they are only needed at the next tile's barrier)
                constexpr int FLY = DSPREAD ? (NI < RB * PER ? NI : RB * PER) : 0;
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0x0f70 | (FLY & 15) | ((FLY >> 4) << 14));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (i >= RB - NIT) store_item(i - (RB - NIT), nxt);
            if constexpr (DSPREAD) {
                if (i * PER < NI) {
                    if (i == 0) dma_part(kn, nxt, std::integral_constant<int, 0>{}, std::integral_constant<int, PER>{});
                    if (i == 1) dma_part(kn, nxt, std::integral_constant<int, PER>{}, std::integral_constant<int, 2 * PER>{});
                    if (i == 2) dma_part(kn, nxt, std::integral_constant<int, 2 * PER>{}, std::integral_constant<int, 3 * PER>{});
                    if (i == 3) dma_part(kn, nxt, std::integral_constant<int, 3 * PER>{}, std::integral_constant<int, 4 * PER>{});
                    if (i == 4) dma_part(kn, nxt, std::integral_constant<int, 4 * PER>{}, std::integral_constant<int, 5 * PER>{});
                }
            }
            mma_block(wf, xf[i & 1][0], xf[i & 1][1], xf[i & 1][2], acc[i]);
            // MFMA first, then the fragment reads of the next block, then the split (2 VALU per MFMA) and its 3 LDS writes
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i + 1 < RB) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            if constexpr (DSPREAD) {
                if (i * PER < NI) {
#pragma unroll
                    for (int q = 0; q < PER; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
            }
            if (i >= RB - NIT) {
#pragma unroll
                for (int q = 0; q < NMMA - 2; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, (44 + NMMA - 3) / (NMMA - 2), 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);

    // epilogue: lane = token (block i, row l15) x 4 consecutive channels per MFMA block j; wave = scene wm x group wn
    const int cbase = col0 + wn * 64 + 4 * g;
    const int srow = wm * ntok;                          // first row of this wave's scene, relative to row0
    bool valid[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) valid[i] = i * 16 + l15 < ntok && srow + i * 16 + l15 < rows_here;
    float* const ob = p.y + (int64_t)z * p.sy + (int64_t)(row0 + srow + l15) * p.ldy + cbase;
    const float* const rbp = p.residual ? p.residual + (int64_t)z * p.sres + (int64_t)(row0 + srow + l15) * p.ldr + cbase : nullptr;
    if constexpr (!GN) {
        // training-step forms (include/diffuscene_hip.h): u also stored to `preact`; or the result multiplied by act_out'(actgrad_x)
        float* const ub = p.preact ? p.preact + (int64_t)z * p.sy + (int64_t)(row0 + srow + l15) * p.ld_preact + cbase : nullptr;
        const float* const gb = p.actgrad_x ? p.actgrad_x + (int64_t)z * p.sy + (int64_t)(row0 + srow + l15) * p.ld_actgrad + cbase : nullptr;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (valid[i]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 y = acc[i][j];
                    if (ub) *reinterpret_cast<f32x4*>(ub + (int64_t)i * 16 * p.ld_preact + j * 16) = y;
                    if (gb) {
                        const f32x4 u = *reinterpret_cast<const f32x4*>(gb + (int64_t)i * 16 * p.ld_actgrad + j * 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] *= dsc_act_grad(u[e], p.act_out);
                    } else if (p.act_out != DSC_ACT_NONE) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = dsc_act(y[e], p.act_out);
                    }
                    if (rbp) y += *reinterpret_cast<const f32x4*>(rbp + (int64_t)i * 16 * p.ldr + j * 16);
                    *reinterpret_cast<f32x4*>(ob + (int64_t)i * 16 * p.ldy + j * 16) = y;
                }
            }
        }
    } else {
        // The wave tile IS one GroupNorm cell: the ntok tokens of scene (row0 / ntok + wm) x the 64 channels of group
        // (col0 / 64 + wn).  Statistics are wave-local (two passes over the lane's accumulators + a wave sum; padding rows
        // masked): no LDS, no block barrier.
        const int scene = min(row0 / ntok + wm, p.m / ntok - 1);
        const bool per_row = p.ss_mode == DSC_SS_PER_TOKEN || p.ss_mode == DSC_SS_PER_SLOT;   // applied per element below
        f32x4 ga[4], be[4], sc[4], sh[4];
        int64_t ssrow = scene;
        if (p.ss_mode == DSC_SS_BY_INDEX) ssrow = dsc_clamp_index(p.ss_index[scene], p.ss_rows);
#pragma unroll
        for (int j = 0; j < 4; ++j) {                    // issued first: their latency hides under the statistics
            ga[j] = *reinterpret_cast<const f32x4*>(p.gamma + cbase + j * 16);
            be[j] = *reinterpret_cast<const f32x4*>(p.beta + cbase + j * 16);
            if (p.ss_mode == DSC_SS_PER_SCENE || p.ss_mode == DSC_SS_BY_INDEX) {
                const float* ssr = p.scale_shift + ssrow * p.ld_ss + cbase + j * 16;
                sc[j] = *reinterpret_cast<const f32x4*>(ssr);
                sh[j] = *reinterpret_cast<const f32x4*>(ssr + p.n);
            } else {
                sc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                sh[j] = sc[j];
            }
        }
        if (p.preact) {
            float* const pb = p.preact + (int64_t)(row0 + srow + l15) * p.ld_preact + cbase;
#pragma unroll
            for (int i = 0; i < RB; ++i)
                if (valid[i]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(pb + (int64_t)i * 16 * p.ld_preact + j * 16) = acc[i][j];
                }
        }
        const float inv_cnt = 1.f / (float)(ntok * 64);
        float s0 = 0.f;
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                s0 += valid[i] ? (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]) : 0.f;
        const float mean = wave_sum_dpp(s0) * inv_cnt;
        float q0 = 0.f;
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = valid[i] ? acc[i][j][e] - mean : 0.f;
                    q0 = fmaf(d, d, q0);
                }
        const float rstd = 1.f / sqrtf(wave_sum_dpp(q0) * inv_cnt + p.eps);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {                // y = z * A + B with A = rstd*gamma*(scale+1), B = (beta - mean*rstd*gamma)*(scale+1) + shift
                const float a = rstd * ga[j][e], sp1 = sc[j][e] + 1.f;
                ga[j][e] = a * sp1;
                be[j][e] = fmaf(be[j][e] - mean * a, sp1, sh[j][e]);
            }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (valid[i]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 y, s1 = {1.f, 1.f, 1.f, 1.f}, s2 = {0.f, 0.f, 0.f, 0.f};
                    if (per_row) {
                        const int64_t row = p.ss_mode == DSC_SS_PER_TOKEN ? (int64_t)row0 + srow + i * 16 + l15 : (int64_t)i * 16 + l15;
                        const float* ssr = p.scale_shift + row * p.ld_ss + cbase + j * 16;
                        s1 = *reinterpret_cast<const f32x4*>(ssr) + 1.f;
                        s2 = *reinterpret_cast<const f32x4*>(ssr + p.n);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = fmaf(acc[i][j][e], ga[j][e], be[j][e]);
                        if (per_row) t = fmaf(t, s1[e], s2[e]);
                        y[e] = dsc_silu_fast(t);
                    }
                    if (rbp) y += *reinterpret_cast<const f32x4*>(rbp + (int64_t)i * 16 * p.ldr + j * 16);
                    *reinterpret_cast<f32x4*>(ob + (int64_t)i * 16 * p.ldy + j * 16) = y;
                }
            }
        }
    }
}

// WM x WN waves (8: one block per CU owns the LDS; 4: half the channels per block for launches that would otherwise leave CUs idle)
template <bool GN, int WM, int WN, int RB, bool DSPREAD = DSC_SPLIT_DSPREAD>
__global__ __launch_bounds__(64 * WM * WN, 1) void gemm_split_kernel(const dsc_gemm_args p, const int ntok) {
    __shared__ __attribute__((aligned(16))) char smem[split_smem_bytes<WM, WN, RB>()];
    gemm_split_tile<GN, WM, WN, RB, DSPREAD>(p, ntok, blockIdx.x, blockIdx.y, smem);
}

template <bool GN, int WM, int WN, int RB>
int launch(const dsc_gemm_args* a, int ntok, hipStream_t s) {
    constexpr int BN = 64 * WN;
    const int scenes = (a->m + ntok - 1) / ntok;
    const unsigned grid = (unsigned)(((scenes + WM - 1) / WM) * (a->n / BN));
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL((gemm_split_kernel<GN, WM, WN, RB>), dim3(grid, (unsigned)a->batch), dim3(64 * WM * WN), 0, s, *a, ntok);
    DSC_LAUNCH_CHECK();
    return 0;
}

}  // namespace dsc_split

// Which split kernel takes this launch: a tile id, or -1 = none (no planes, DSC_GEMM=f32 in the environment, or a shape / alignment /
// launch size this path does not cover: the exact-f32 MFMA kernel runs).
enum { T_GN_32 = 0, T_GN_80_W8, T_GN_80_W4, T_GN_48, T_GN_64, T_160x256, T_256x128, T_128x128, T_160x128_W4, T_64x256, T_WAVE_GN, T_WAVE_DENSE, T_WAVE_GN_64 };

// The arithmetic of the GEMM entry points -- ONE source of truth for the library and its host code (engine, training plan, bench all
// ask dsc_get_gemm_arithmetic): 1 = split-bf16 wherever a launch qualifies (default), 0 = exact-f32 MFMA everywhere.  Initial value
// from the environment, strictly: DSC_GEMM unset / "" / "split" -> 1, "f32" -> 0, anything else -> DSC_EINVAL (every GEMM launch
// then fails with DSC_EINVAL instead of silently picking one).  dsc_set_gemm_arithmetic switches it per call (process-wide; launches
// already captured in a hipGraph keep the kernels they were captured with).
static std::atomic<int> g_gemm_arith{-1000};

extern "C" int dsc_get_gemm_arithmetic(void) {
    int m = g_gemm_arith.load(std::memory_order_relaxed);
    if (m == -1000) {
        const char* e = getenv("DSC_GEMM");
        m = (!e || !e[0] || !strcmp(e, "split")) ? 1 : !strcmp(e, "f32") ? 0 : DSC_EINVAL;
        g_gemm_arith.store(m, std::memory_order_relaxed);
    }
    return m;
}

extern "C" int dsc_set_gemm_arithmetic(int32_t mode) {
    if (mode != 0 && mode != 1) return DSC_EINVAL;
    g_gemm_arith.store(mode, std::memory_order_relaxed);
    return 0;
}

// The wave-autonomous family (gemm_split_wave.h), 1 = wherever a launch qualifies, 0 = never; DSC_WAVE in the environment, strictly.
static std::atomic<int> g_split_wave{-1000};

extern "C" int dsc_get_split_wave(void) {
    int m = g_split_wave.load(std::memory_order_relaxed);
    if (m == -1000) {
        const char* e = getenv("DSC_WAVE");
        m = (!e || !e[0] || !strcmp(e, "1") || !strcmp(e, "auto")) ? 1 : !strcmp(e, "0") ? 0 : DSC_EINVAL;
        g_split_wave.store(m, std::memory_order_relaxed);
    }
    return m;
}

extern "C" int dsc_set_split_wave(int32_t mode) {
    if (mode != 0 && mode != 1) return DSC_EINVAL;
    g_split_wave.store(mode, std::memory_order_relaxed);
    return 0;
}

// Does the wave-autonomous kernel take this launch?  One wave = one scene (GroupNorm form: 17..80 tokens) or one group of 80 dense rows
// x 128 channels, one wave per SIMD: the launch must come out at a whole number of waves per SIMD, nearly (>= 80 % of the chip in every
// round; measured, profiles/r06_plan_profile_sample_wave{0,1}.txt: n = 384 at M = 20480 = 768 waves, three per CU, 45.8 vs 42.7 us on the
// block-staged 256 x 128 tile), and K >= 256 (a four-tile K loop does not pay for the longer prologue: 29.7 vs 28.2 us).
static int select_wave(const dsc_gemm_args* a, bool gn) {
    if (dsc_get_split_wave() != 1) return -1;
    const int K = a->k1 + a->k2;
    if ((a->n % 128) || (a->k1 % 64) || (a->k2 % 64) || K < 256) return -1;
    if (a->batch != 1 && (gn || a->sw != (int64_t)a->n * K || a->ldw != K || (a->sy & 3) || (a->sres & 3) || (a->sbias & 3))) return -1;
    if (!dsc_aligned16(a->y) || (a->ldy & 3) || a->ldy < a->n) return -1;
    if (a->bias && !dsc_aligned16(a->bias)) return -1;
    if (a->residual && (!dsc_aligned16(a->residual) || (a->ldr & 3) || a->ldr < a->n)) return -1;
    if (a->preact && (!dsc_aligned16(a->preact) || (a->ld_preact & 3) || a->ld_preact < a->n || a->batch != 1)) return -1;
    if (a->actgrad_x && (gn || !dsc_aligned16(a->actgrad_x) || (a->ld_actgrad & 3) || a->ld_actgrad < a->n || a->batch != 1)) return -1;
    if (3LL * a->batch * a->n * K * 2 >= 0x7fffffffLL) return -1;                       // 32-bit offsets into the planes
    const int64_t ld_max = a->lda1 > a->lda2 ? a->lda1 : a->lda2;
    if (ld_max * 4 * 96 >= 0x7fffffffLL) return -1;                                     // 32-bit byte offsets inside a wave's rows
    long waves;
    if (a->gnb_z) {
        // GroupNorm-backward epilogue: dense product over whole scenes of 65..80 tokens, the epilogue's operands in whole 16-byte columns
        const int N = a->tokens_per_scene;
        if (gn || a->batch != 1 || N <= 64 || N > 80 || (a->m % N) || a->bias || a->residual || a->preact || a->actgrad_x || a->act_out != DSC_ACT_NONE) return -1;
        if (!a->gamma || !a->beta || !a->gnb_dgamma || !a->gnb_dbeta || !a->gnb_dbias || (a->ss_mode != DSC_SS_NONE && a->ss_mode != DSC_SS_PER_SCENE)) return -1;
        if (!dsc_aligned16(a->gnb_z) || (a->ld_gnb_z & 3) || a->ld_gnb_z < a->n || !dsc_aligned16(a->gamma) || !dsc_aligned16(a->beta)) return -1;
        if (!dsc_aligned16(a->gnb_dgamma) || !dsc_aligned16(a->gnb_dbeta) || !dsc_aligned16(a->gnb_dbias) || (a->gnb_pstride & 3)) return -1;
        if (a->ss_mode == DSC_SS_PER_SCENE && (!a->scale_shift || !dsc_aligned16(a->scale_shift) || (a->ld_ss & 3))) return -1;
        if (a->gnb_dss && (!dsc_aligned16(a->gnb_dss) || (a->ld_gnb_dss & 3))) return -1;
        waves = (long)(a->m / N) * (a->n / 128);
        const long r = (waves + 1023) / 1024;
        return waves * 5 >= r * 1024 * 4 ? T_WAVE_DENSE : -1;
    }
    if (gn) {
        const int N = a->tokens_per_scene;
        if (N <= 16 || N > 80) return -1;
        const bool perrow = a->ss_mode == DSC_SS_PER_TOKEN || a->ss_mode == DSC_SS_PER_SLOT;
        if (perrow && a->ld_ss < 2 * (int64_t)a->n) return -1;
        waves = (long)(a->m / N) * (a->n / 128);
    } else {
        waves = (long)((a->m + 79) / 80) * (a->n / 128) * a->batch;
    }
    const long rounds = (waves + 1023) / 1024;
    if (waves * 5 >= rounds * 1024 * 4) return gn ? T_WAVE_GN : T_WAVE_DENSE;
    // half-size GroupNorm launches of 65..80-token scenes (B = 128: 512 waves of 80 x 128 would leave half the SIMDs idle): waves of 80 x 64,
    // one GroupNorm cell each.  Measured against the four-wave block-staged tile at B = 128 (profiles/r06_wave_probe_j4.txt): 33.7 vs 38.0 us,
    // 38.4 vs 47.4 with a residual, 34.6 vs 41.6 with per-slot (scale, shift); at two rounds (B = 256) the 80 x 128 wave tile wins (57.5 vs 67.7).
    if (gn && a->tokens_per_scene > 64) {
        const long w4 = 2 * waves, r4 = (w4 + 1023) / 1024;
        if (r4 == 1 && w4 * 5 >= 1024 * 4) return T_WAVE_GN_64;
    }
    return -1;
}

static int select_block_tile(const dsc_gemm_args* a, bool gn);

// have_planes: the planes question is asked for a launch as it is (dispatch) or for a launch whose planes do not exist yet
static int select_tile(const dsc_gemm_args* a, bool gn, bool assume_planes = false) {
    if (dsc_get_gemm_arithmetic() != 1 || (!assume_planes && !a->w_planes)) return -1;
    const int w = select_wave(a, gn);
    return w >= 0 ? w : select_block_tile(a, gn);
}

static int select_block_tile(const dsc_gemm_args* a, bool gn) {
    if (a->gnb_z) return -1;                             // (the GroupNorm-backward epilogue exists on the wave-autonomous kernel only)
    const int K = a->k1 + a->k2;
    // grouped launches: the weights of the problems must be the row blocks of one stacked matrix (planes [3][batch n][K])
    if (a->batch != 1 && (gn || a->sw != (int64_t)a->n * K || a->ldw != K || (a->sy & 3) || (a->sres & 3) || (a->sbias & 3))) return -1;
    if ((a->n % 128) || (K % 32)) return -1;
    if (!dsc_aligned16(a->w_planes) || !dsc_aligned16(a->y) || (a->ldy & 3)) return -1;
    if (a->bias && !dsc_aligned16(a->bias)) return -1;
    if (a->residual && (!dsc_aligned16(a->residual) || (a->ldr & 3))) return -1;
    if (!gn && a->preact && (!dsc_aligned16(a->preact) || (a->ld_preact & 3) || a->batch != 1)) return -1;
    if (a->actgrad_x && (gn || !dsc_aligned16(a->actgrad_x) || (a->ld_actgrad & 3) || a->batch != 1)) return -1;
    if (3LL * a->batch * a->n * K * 2 >= 0x7fffffffLL) return -1;                       // 32-bit DMA offsets into the planes
    const int64_t ld_max = a->lda1 > a->lda2 ? a->lda1 : a->lda2;
    if (ld_max * 4 * 320 >= 0x7fffffffLL) return -1;                                    // 32-bit byte offsets inside a token tile
    const bool wide = (a->n % 256) == 0;
    // A launch must fill the chip: one block per CU (two stages of operand planes own the LDS), so fewer than ~160 blocks leave more
    // than a third of the 256 CUs idle and the f32-MFMA kernel's smaller tiles win (measured: text / M = 1536 2.6 -> 3.3 ms per
    // step on this path, M = 10240 with 128 eight-wave blocks 51 -> 58 us per launch).  4-wave blocks (half the channels) double the
    // block count for the half-size launches.
    constexpr long MIN_BLOCKS = 160;
    if (gn) {
        const int N = a->tokens_per_scene;
        if (N <= 16 || N > 80) return -1;            // (N <= 16: one-row-block tiles measured and rejected, profiles/r04_small_tiles.txt)
        const long S = a->m / N;
        if (N <= 32) return ((S + 3) / 4) * (a->n / 128) >= MIN_BLOCKS ? T_GN_32 : -1;
        const long b8 = wide ? ((S + 1) / 2) * (a->n / 256) : 0, b4 = ((S + 1) / 2) * (a->n / 128);
        if (N > 64) return b8 >= MIN_BLOCKS + 32 ? T_GN_80_W8 : b4 >= MIN_BLOCKS ? T_GN_80_W4 : -1;
        if (b8 < MIN_BLOCKS) return -1;
        return N <= 48 ? T_GN_48 : T_GN_64;
    }
    // dense rows: tile = (16 RB WM) x (64 WN); fewest rounds of 256 CUs x tile area wins, ties to the earlier (larger) tile
    struct Cand { int bm, bn, id; };
    const Cand cands[5] = {{160, 256, T_160x256}, {256, 128, T_256x128}, {160, 128, T_160x128_W4}, {128, 128, T_128x128}, {64, 256, T_64x256}};
    int best = -1;
    long best_cost = 0, best_blk = 0;
    for (int i = 0; i < 5; ++i) {
        if (cands[i].bn == 256 && !wide) continue;
        const long nblk = (long)((a->m + cands[i].bm - 1) / cands[i].bm) * (a->n / cands[i].bn) * a->batch;
        const long c = ((nblk + 255) / 256) * (long)cands[i].bm * cands[i].bn;
        if (best < 0 || c < best_cost) { best = cands[i].id; best_cost = c; best_blk = nblk; }
    }
    return best_blk >= MIN_BLOCKS ? best : -1;
}

// -> 0 / error code when the split-bf16 path took the launch, DSC_SPLIT_NOT_TAKEN when the caller should run the f32-MFMA kernel
int dsc_gemm_try_split(const dsc_gemm_args* a, bool gn, hipStream_t s) {
    using namespace dsc_split;
    if (dsc_get_gemm_arithmetic() < 0 || dsc_get_split_wave() < 0) return DSC_EINVAL;   // DSC_GEMM / DSC_WAVE hold an unknown value
    const int N = a->tokens_per_scene;
    const int tile = select_tile(a, gn);
    if (tile < 0) return DSC_SPLIT_NOT_TAKEN;
    // the planes must have the layout the chosen kernel reads (dsc_gemm_planes_layout said which before they were made)
    const int want = (tile == T_WAVE_GN || tile == T_WAVE_DENSE || tile == T_WAVE_GN_64) ? DSC_PLANES_FRAGMENT : DSC_PLANES_ROWMAJOR;
    if (a->w_planes_layout != want) return DSC_EINVAL;
    switch (tile) {
        case T_GN_32: return launch<true, 4, 2, 2>(a, N, s);
        case T_GN_80_W8: return launch<true, 2, 4, 5>(a, N, s);
        case T_GN_80_W4: return launch<true, 2, 2, 5>(a, N, s);
        case T_GN_48: return launch<true, 2, 4, 3>(a, N, s);
        case T_GN_64: return launch<true, 2, 4, 4>(a, N, s);
        case T_160x256: return launch<false, 2, 4, 5>(a, 80, s);
        case T_256x128: return launch<false, 4, 2, 4>(a, 64, s);
        case T_128x128: return launch<false, 4, 2, 2>(a, 32, s);
        case T_160x128_W4: return launch<false, 2, 2, 5>(a, 80, s);
        case T_64x256: return launch<false, 2, 4, 2>(a, 32, s);
        case T_WAVE_DENSE:
            if (a->gnb_z) return dsc_wave::launch<false, 5, false, 8, true>(a, a->w_planes, N, s);       // GroupNorm-backward epilogue: rows = scenes
            return dsc_wave::launch<false, 5>(a, a->w_planes, 80, s);
        case T_WAVE_GN_64:
            return (a->ss_mode == DSC_SS_PER_TOKEN || a->ss_mode == DSC_SS_PER_SLOT) ? dsc_wave::launch<true, 5, true, 4>(a, a->w_planes, N, s)
                                                                                     : dsc_wave::launch<true, 5, false, 4>(a, a->w_planes, N, s);
        case T_WAVE_GN: {
            const bool perrow = a->ss_mode == DSC_SS_PER_TOKEN || a->ss_mode == DSC_SS_PER_SLOT;
            const int rb = (N + 15) / 16;
            if (perrow) {
                return rb == 5 ? dsc_wave::launch<true, 5, true>(a, a->w_planes, N, s) : rb == 4 ? dsc_wave::launch<true, 4, true>(a, a->w_planes, N, s)
                     : rb == 3 ? dsc_wave::launch<true, 3, true>(a, a->w_planes, N, s) : dsc_wave::launch<true, 2, true>(a, a->w_planes, N, s);
            }
            return rb == 5 ? dsc_wave::launch<true, 5>(a, a->w_planes, N, s) : rb == 4 ? dsc_wave::launch<true, 4>(a, a->w_planes, N, s)
                 : rb == 3 ? dsc_wave::launch<true, 3>(a, a->w_planes, N, s) : dsc_wave::launch<true, 2>(a, a->w_planes, N, s);
        }
        default: return DSC_SPLIT_NOT_TAKEN;
    }
}

// 1: dsc_gemm_f32 (gn = 0) / dsc_gemm_gn_silu_f32 (gn != 0) would run this launch on the split-bf16 kernel; 0: exact-f32 MFMA
extern "C" int dsc_gemm_arithmetic(const dsc_gemm_args* a, int32_t gn) {
    if (!a || a->m <= 0 || a->n <= 0 || a->k1 <= 0) return DSC_EINVAL;
    return select_tile(a, gn != 0) >= 0 ? 1 : 0;
}

// The tile class (DSC_TILE_* of the header; the enum above) the split-bf16 path would run this launch on, or -1
extern "C" int dsc_gemm_split_tile(const dsc_gemm_args* a, int32_t gn) {
    if (!a || a->m <= 0 || a->n <= 0 || a->k1 <= 0) return DSC_EINVAL;
    return select_tile(a, gn != 0);
}

extern "C" int dsc_gemm_planes_layout(const dsc_gemm_args* a, int32_t gn) {
    if (!a || a->m <= 0 || a->n <= 0 || a->k1 <= 0) return DSC_EINVAL;
    const int tile = select_tile(a, gn != 0, true);
    return tile < 0 ? -1 : (tile == T_WAVE_GN || tile == T_WAVE_DENSE || tile == T_WAVE_GN_64) ? DSC_PLANES_FRAGMENT : DSC_PLANES_ROWMAJOR;
}

extern "C" int dsc_split_bf16x3_f32(const dsc_split_item* items, int32_t count, dsc_stream_t stream) {
    using namespace dsc_split;
    if (!items || count < 1 || count > DSC_WS_MAX) return DSC_EINVAL;
    SplitBatch b;
    long max_items = 0;
    for (int i = 0; i < count; ++i) {
        const dsc_split_item& it = items[i];
        if (!it.w || !it.planes || it.rows <= 0 || it.cols <= 0) return DSC_EINVAL;
        if (it.transpose & ~(DSC_SPLIT_TRANSPOSE | DSC_SPLIT_FRAGMENT)) return DSC_EINVAL;
        const bool tr = (it.transpose & DSC_SPLIT_TRANSPOSE) != 0;
        const int co = tr ? it.rows : it.cols, ro = tr ? it.cols : it.rows;
        if (co % 8) return DSC_EINVAL;
        if ((it.transpose & DSC_SPLIT_FRAGMENT) && ((ro % 16) || (co % 32))) return DSC_EINVAL;
        if (!dsc_aligned16(it.planes) || (!tr && (!dsc_aligned16(it.w) || (it.ldw & 3)))) return DSC_EALIGN;
        b.it[i] = it;
        const long n = (long)it.rows * it.cols / 8;
        if (n > max_items) max_items = n;
    }
    long bx = (max_items + 255) / 256;
    if (bx > 1024) bx = 1024;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)bx, (unsigned)count), dim3(256), 0, static_cast<hipStream_t>(stream), b);
    DSC_LAUNCH_CHECK();
    return 0;
}

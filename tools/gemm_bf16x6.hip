// EXPERIMENT (not part of the product, not linked into libdiffuscene_hip.so): an f32-accurate GEMM on the bf16 matrix cores.
//
//   out[m][n] = bias[n] + sum_k x[m][k] * w[n][k]          x, w, out: f32 in HBM, as in dsc_gemm_f32
//
// Every f32 operand is split EXACTLY into three bf16 pieces (x = x1 + x2 + x3, round-to-nearest at each step, 8 + 8 + 8
// mantissa bits), every bf16 x bf16 product is exact in f32, and the six products whose weight is >= 2^-18 of the leading
// one are accumulated in f32 by v_mfma_f32_16x16x32_bf16:
//
//   x*w ~= x1*w1 + x1*w2 + x2*w1 + x2*w2 + x1*w3 + x3*w1       (dropped: x2*w3, x3*w2, x3*w3 <= 2^-26 relative)
//
// tools/gemm_bf16x6.py measures the error against an f64 product next to dsc_gemm_f32's (CPU emulation in
// profiles/r02_bf16x6_numerics.txt: rms error 1.2e-7 vs 2.9e-7 for the f32 MFMA order of summation).  The bf16 pipe is 16x
// the f32 MFMA rate, so six products are 2.67x the f32-MFMA roofline -- IF the operand split (11 VALU per pair of x
// elements, done on the fragments after the LDS read) and the LDS traffic fit under the MFMAs.  That is what this measures.
//
// Layout: 8 waves as WM x WN; a wave owns ONE scene (ntok <= 16*RB tokens, padded to RB MFMA row blocks inside LDS) x 64
// channels = RB x 4 MFMA blocks of 16 x 16 -- exactly one GroupNorm cell.  RB = 5: 80-token scenes, block 160 tokens x 256
// channels (M = 20480, n = 512 -> 256 blocks = one round of 256 CUs) or 320 x 128; RB = 2: scenes of <= 32 tokens (N = 21:
// 4 scenes x 128 channels per block -> 256 blocks at B = 256).  Plain GEMMs use ntok = 16*RB (dense).  BK = 32 = one MFMA
// k-step.  The weights are pre-split into three bf16 planes [3][n][k] by split_planes_kernel (once per weight update) and
// staged as planes by LDS-DMA (lane-linear image, XOR swizzle applied on the global side).  The tokens, per main-loop form:
//   PIPE 0-2: x stays f32 in LDS (LDS-DMA as well) and every wave splits its fragments after the LDS read -- compiler-
//             scheduled (0), split of block i+1 slotted between the MFMAs of block i inside a K tile (1), or across tiles (2);
//   PIPE 3:   the staging threads load f32 rows into registers, split ONCE per block under the previous tile's MFMAs and write
//             three bf16 planes to LDS (no wave repeats the split of rows it shares; fragments are three 16-byte reads).
// The MFMA computes out^T (weights as the row operand) so that each lane holds 4 consecutive channels of one token: the
// plain epilogue is one 16-byte store per accumulator; the fused Block epilogue (all conditioning modes of the product's
// dsc_gemm_gn_silu_f32, optional saved pre-activation) needs no LDS and no block barrier.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 32, NW = 8, T = NW * 64;
#ifdef RES_WARM
constexpr int WARM_TILES = 4;
#endif
// A block is WM x WN waves of 80 tokens x 64 channels (8 waves): 2 x 4 -> 160 x 256 (one round at M = 20480, n = 512),
// 4 x 2 -> 320 x 128 (n = 384 / 128 and other multiples of 128).
template <int WM, int WN, int RB>
struct Cfg {
    static_assert(WM * WN == NW, "8 waves");
    static constexpr int BM = 16 * RB * WM, BN = 64 * WN;        // BM = LDS rows (scenes padded to 16*RB); global rows = WM * ntok
    static constexpr int A_STAGE = BM * BK * 4;                  // f32 tokens
    static constexpr int B_PLANE = BN * BK * 2;                  // one bf16 weight plane
    static constexpr int STAGE = A_STAGE + 3 * B_PLANE;
    static constexpr int CH_A = A_STAGE / 1024, CH_PL = B_PLANE / 1024, CH = CH_A + 3 * CH_PL;   // one-KiB wave transfers
    static constexpr int NI = (CH + NW - 1) / NW;                // per wave (a ragged last round re-fetches chunks 0..: same bytes)
    static constexpr bool P3 = 2 * (3 * BM * BK * 2 + 3 * B_PLANE) <= 160 * 1024;   // PIPE 3 (x staged as three bf16 planes) fits two stages
};

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    f32x2v v = {a, b};
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

__global__ void split_planes_kernel(const float* __restrict__ w, long count, uint16_t* __restrict__ planes) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= count) return;
    const float x0 = w[i], x1 = i + 1 < count ? w[i + 1] : 0.f;
    const unsigned u1 = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - bf_lo(u1), r1 = x1 - bf_hi(u1);
    const unsigned u2 = cvt_pk_bf16(r0, r1);
    const unsigned u3 = cvt_pk_bf16(r0 - bf_lo(u2), r1 - bf_hi(u2));
    const unsigned u[3] = {u1, u2, u3};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        planes[p * count + i] = (uint16_t)(u[p] & 0xffffu);
        if (i + 1 < count) planes[p * count + i + 1] = (uint16_t)(u[p] >> 16);
    }
}

struct Args {
    const float* x;          // [m][lda] f32: the first k1 columns of the K range
    const float* x2;         // [m][lda] f32: columns k1 .. k (torch.cat of a skip connection), or null when k1 == k
    int k1;
    const uint16_t* planes;  // [3][n][k] bf16
    const float* bias;       // [n] or null
    float* out;              // [m][ldc]
    int m, n, k, lda, ldc;
    int ntok;                // tokens per scene (<= 16*RB); dense GEMM: 16*RB
    // GroupNorm(8 groups of 64 channels over the ntok tokens of a scene) + (scale + 1, shift) + SiLU (+ residual) epilogue
    const float* gamma; const float* beta; float eps;
    const float* scale_shift; int ld_ss;   // rows [scale(n) | shift(n)], or null
    int ss_mode;                           // the product's DSC_SS_*: 0 none, 1 row = token, 2 row = scene, 3 row = token % ntok, 4 row = ss_index[scene]
    const int64_t* ss_index;               // mode 4: per scene (the timestep vector while sampling)
    float* preact; int ld_pre;             // optional: z = x.w^T + bias saved for the backward
    const float* residual; int ldr;        // or null (added after the activation)
    int act;                               // plain epilogue: 0 none, 1 GELU (erf), 2 SiLU
};

template <int PRODUCTS, int PIPE, bool GN = false, int WM = 2, int WN = 4, int RB = 5>
__global__ __launch_bounds__(T, 1) void gemm_bf16_split_kernel(const Args p) {
    static_assert(PIPE != 2 || RB >= 2, "the cross-tile pipeline reads two blocks ahead: one-block scenes would need a third stage");
    using C = Cfg<WM, WN, RB>;
    constexpr int BM = C::BM, BN = C::BN, B_PLANE = C::B_PLANE, CH_PL = C::CH_PL;
    // per-stage LDS: [x region | 3 weight planes].  PIPE 0-2: x is the f32 tile, filled by DMA.  PIPE 3: x is three bf16
    // planes [3][BM][32] written by the staging threads (split once per block), only the weights come by DMA.
    constexpr int X_PLANE = BM * BK * 2;
    constexpr int XA = PIPE == 3 ? 3 * X_PLANE : C::A_STAGE;
    constexpr int STAGE = XA + 3 * B_PLANE;
    constexpr int CH_A = PIPE == 3 ? 0 : C::CH_A, CH = CH_A + 3 * CH_PL, NI = (CH + NW - 1) / NW;
    constexpr int DUMP = (PIPE == 3 && (BM * 4 / NW) % 64) ? 1024 : 0;     // PIPE 3: where the idle lanes of a ragged item round write
    static_assert(2 * STAGE + DUMP <= 160 * 1024, "two stages must fit the 160 KiB LDS");
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE + DUMP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave_u % WM, wn = wave_u / WM;
    // XCD-aware block order: the two channel halves of a token block sit next to each other on one XCD (shared x rows in its L2)
    const int scenes = (p.m + p.ntok - 1) / p.ntok;
    const int cbs = p.n / BN, rbs = (scenes + WM - 1) / WM;
    int rb, cb;
    if ((rbs & 7) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        rb = xcd * (rbs >> 3) + idx / cbs;
        cb = idx % cbs;
    } else {
        rb = blockIdx.x / cbs;
        cb = blockIdx.x % cbs;
    }
    const int row0 = rb * WM * p.ntok, col0 = cb * BN;          // first global token row of the block
    const int rows_here = p.m - row0;                            // valid global rows from row0 on (may exceed the block)
    const float* const xb = p.x + (int64_t)row0 * p.lda;
    const float* const xb2 = p.x2 ? p.x2 + (int64_t)row0 * p.lda : xb;
    const uint16_t* const wb = p.planes + (int64_t)col0 * p.k;
    const int plane_bytes = p.n * p.k * 2;

    int dvoff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int c = wave_u + NW * i;
        if (c >= CH) c -= CH;
        if (c < CH_A) {                                  // 8 LDS rows x 128 B; k-quad q lands in slot q ^ (row & 7)
            const int r = c * 8 + (lane >> 3);           // LDS row: scene r / (16*RB), token r % (16*RB); padding rows re-fetch row 0
            const int sc = r / (16 * RB), tk = r % (16 * RB), gr = sc * p.ntok + tk;
            dvoff[i] = (tk < p.ntok && gr < rows_here ? gr : 0) * p.lda * 4 + (((lane & 7) ^ (r & 7)) << 4);
        } else {                                         // 16 channel rows x 64 B of one plane; k-octet g lands in slot g ^ ((n >> 1) & 3)
            const int cbk = c - CH_A, plane = cbk / CH_PL, nrow = (cbk % CH_PL) * 16 + (lane >> 2);
            dvoff[i] = plane * plane_bytes + nrow * p.k * 2 + (((lane & 3) ^ ((nrow >> 1) & 3)) << 4);
        }
    }
    auto dma_tile = [&](int kt, char* stage) {
        const int k0 = kt * BK;
        const bool seg1 = k0 < p.k1;
        const float* const xs_ = seg1 ? xb : xb2;
        const int sx = (seg1 ? k0 : k0 - p.k1) * 4;
        __attribute__((address_space(3))) char* lbase = (__attribute__((address_space(3))) char*)stage;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int c = wave_u + NW * i;
            if (c >= CH) c -= CH;
            const bool isx = c < CH_A;                   // wave-uniform
#if defined(__HIP_DEVICE_COMPILE__)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                isx ? (void*)const_cast<float*>(xs_) : (void*)const_cast<uint16_t*>(wb), 0, 0x7fffffff, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lbase + (isx ? c * 1024 : XA + (c - CH_A) * 1024), 16, dvoff[i], isx ? sx : k0 * 2, 0, 0);
#else
            (void)isx; (void)sx; (void)xs_; (void)lbase;
#endif
        }
    };

    // accumulators: out^T blocks, lane = (token lane&15, channels 4*(lane>>4) .. +3); bias folded into the initial value
    f32x4 acc[RB][4];
    const int g = lane >> 4, l15 = lane & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) b4 = *(const f32x4*)(p.bias + col0 + wn * 64 + j * 16 + 4 * g);
#pragma unroll
        for (int i = 0; i < RB; ++i) acc[i][j] = b4;
    }

    // per-lane LDS byte offsets of the fragments inside a stage
    int aoff[RB][2], woff[4];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int r = (wm * RB + i) * 16 + l15;
        aoff[i][0] = r * 128 + (((2 * g) ^ (r & 7)) << 4);
        aoff[i][1] = r * 128 + (((2 * g + 1) ^ (r & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nr = wn * 64 + j * 16 + l15;
        woff[j] = XA + nr * 64 + ((g ^ ((nr >> 1) & 3)) << 4);
    }

    const int KT = p.k / BK;
    auto mma_block = [&](const bf16x8 (&wf)[4][3], const bf16x8& x1, const bf16x8& x2, const bf16x8& x3, f32x4 (&c)[4]) {
        // product-major: the same accumulator comes round every 4th MFMA (64 cycles apart); small terms first
        if (PRODUCTS >= 6) {
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][2], x1, c[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], x3, c[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][1], x2, c[j], 0, 0, 0);
        }
        if (PRODUCTS >= 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][1], x1, c[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], x2, c[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], x1, c[j], 0, 0, 0);
    };
    constexpr int NMMA = 4 * PRODUCTS;                   // MFMAs per 16-token block
    dma_tile(0, smem);
    if constexpr (PIPE == 0) {
        // compiler-scheduled: split and MFMA stretches alternate; the SIMD's other wave is what fills the gaps
        for (int kt = 0; kt < KT; ++kt) {
            __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): this wave's chunks of tile kt are in LDS
            __syncthreads();                             // everyone's are, and nobody still reads the other stage
            if (kt + 1 < KT) dma_tile(kt + 1, smem + ((kt + 1) & 1) * STAGE);
            const char* st = smem + (kt & 1) * STAGE;
            bf16x8 wf[4][3];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wf[j][pl] = *(const bf16x8*)(st + woff[j] + pl * B_PLANE);
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const f32x4 lo = *(const f32x4*)(st + aoff[i][0]), hi = *(const f32x4*)(st + aoff[i][1]);
                bf16x8 x1, x2, x3;
                split8(lo, hi, x1, x2, x3);
                mma_block(wf, x1, x2, x3, acc[i]);
            }
        }
    } else if constexpr (PIPE == 1) {
        // PIPE 1: inside a K tile the split of token block i+1 (44 VALU) is slotted between the MFMAs of block i, and the
        // f32 fragment of block i+2 is read under them; only block 0's split and the weight-plane reads are exposed per tile.
        for (int kt = 0; kt < KT; ++kt) {
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
            if (kt + 1 < KT) dma_tile(kt + 1, smem + ((kt + 1) & 1) * STAGE);
            const char* st = smem + (kt & 1) * STAGE;
            bf16x8 wf[4][3];
            f32x4 raw[2][2];
            bf16x8 xs[2][3];
            raw[0][0] = *(const f32x4*)(st + aoff[0][0]);
            raw[0][1] = *(const f32x4*)(st + aoff[0][1]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wf[j][pl] = *(const bf16x8*)(st + woff[j] + pl * B_PLANE);
            if constexpr (RB > 1) {
                raw[1][0] = *(const f32x4*)(st + aoff[1][0]);
                raw[1][1] = *(const f32x4*)(st + aoff[1][1]);
            }
            split8(raw[0][0], raw[0][1], xs[0][0], xs[0][1], xs[0][2]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                if (i + 2 < RB) {
                    raw[i & 1][0] = *(const f32x4*)(st + aoff[i + 2][0]);
                    raw[i & 1][1] = *(const f32x4*)(st + aoff[i + 2][1]);
                }
                if (i + 1 < RB)
                    split8(raw[(i + 1) & 1][0], raw[(i + 1) & 1][1], xs[(i + 1) & 1][0], xs[(i + 1) & 1][1], xs[(i + 1) & 1][2]);
                mma_block(wf, xs[i & 1][0], xs[i & 1][1], xs[i & 1][2], acc[i]);
                if (i + 1 < RB) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i + 2 < RB) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                    for (int q = 0; q < NMMA - 2; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, (44 + NMMA - 3) / (NMMA - 2), 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if constexpr (PIPE == 2) {
        // PIPE 2: the pipeline runs ACROSS K tiles.  Block s = RB*kt + i: its MFMAs use the split made under block s-1's MFMAs
        // from the f32 fragment read under block s-2's.  Tile kt+1 is confirmed (vmcnt(0) + barrier) at the start of block
        // B0 = RB-2 of tile kt: by then every fragment of tile kt is in registers, so the same barrier frees stage kt&1 for the
        // DMA of tile kt+2, and blocks B0, B0+1 read tile kt+1's first token fragments and its weight planes (second register
        // set).  No exposed split or LDS latency per tile; the tail issues harmless duplicate DMAs / reads instead of branching.
        constexpr int B0 = RB - 2;
        bf16x8 wfA[4][3], wfB[4][3];
        f32x4 raw[2][2];
        bf16x8 xs[2][3];
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        dma_tile(KT > 1 ? 1 : 0, smem + STAGE);
        raw[0][0] = *(const f32x4*)(smem + aoff[0][0]);
        raw[0][1] = *(const f32x4*)(smem + aoff[0][1]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wfA[j][pl] = *(const bf16x8*)(smem + woff[j] + pl * B_PLANE);
        raw[1][0] = *(const f32x4*)(smem + aoff[1][0]);
        raw[1][1] = *(const f32x4*)(smem + aoff[1][1]);
        split8(raw[0][0], raw[0][1], xs[0][0], xs[0][1], xs[0][2]);
        __builtin_amdgcn_sched_barrier(0);
        auto tile = [&](auto par_c, int kt, const bf16x8 (&wfc)[4][3], bf16x8 (&wfn)[4][3]) {
            constexpr int PAR = decltype(par_c)::value;          // kt & 1 == PAR; block i of this tile has parity (PAR*RB + i) & 1
            const char* cur = smem + PAR * STAGE;
            const char* nxt = smem + (PAR ^ 1) * STAGE;
            auto block = [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int sp = (PAR * RB + i) & 1;
                if constexpr (i == B0) {
                    __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0): tile kt+1 landed, my reads of tile kt done
                    __syncthreads();
                    dma_tile(min(kt + 2, KT - 1), const_cast<char*>(cur));
                    __builtin_amdgcn_sched_barrier(0);
                }
                const char* src = i + 2 < RB ? cur : nxt;
                constexpr int blk = i + 2 < RB ? i + 2 : i + 2 - RB;
                raw[sp][0] = *(const f32x4*)(src + aoff[blk][0]);
                raw[sp][1] = *(const f32x4*)(src + aoff[blk][1]);
                if constexpr (i >= B0) {
#pragma unroll
                    for (int j = 2 * (i - B0); j < 2 * (i - B0) + 2; ++j)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) wfn[j][pl] = *(const bf16x8*)(nxt + woff[j] + pl * B_PLANE);
                }
                split8(raw[sp ^ 1][0], raw[sp ^ 1][1], xs[sp ^ 1][0], xs[sp ^ 1][1], xs[sp ^ 1][2]);
                mma_block(wfc, xs[sp][0], xs[sp][1], xs[sp][2], acc[i]);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                if constexpr (i >= B0) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
                }
#pragma unroll
                for (int q = 0; q < NMMA - 2 - (i >= B0 ? 3 : 0); ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, (44 + NMMA - 3) / (NMMA - 2) + (i >= B0 ? 1 : 0), 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            block(std::integral_constant<int, 0>{});
            block(std::integral_constant<int, 1>{});
            if constexpr (RB > 2) block(std::integral_constant<int, 2>{});
            if constexpr (RB > 3) block(std::integral_constant<int, 3>{});
            if constexpr (RB > 4) block(std::integral_constant<int, 4>{});
            static_assert(RB <= 5, "blocks are spelled out up to 5");
        };
        for (int kt = 0; kt < KT; kt += 2) {                     // KT is even (host checks k % 64 == 0)
            tile(std::integral_constant<int, 0>{}, kt, wfA, wfB);
            tile(std::integral_constant<int, 1>{}, kt + 1, wfB, wfA);
        }
    } else {
        // PIPE 3: split ONCE per block.  The staging threads load the f32 token tile of K tile kt+1 into registers (8
        // consecutive k of one token row per item, 1 or 2 items per lane), split it under the MFMAs of tile kt and write three
        // bf16 planes [3][BM][32] into the other LDS stage -- the same row layout and swizzle as the weight planes -- so a token
        // fragment is three 16-byte reads and no wave repeats the split of rows it shares with the other WN-1 waves (44 VALU
        // per item instead of 44 per 16-token block and wave: 88 instead of 220 per wave and tile at 160 x 256).
        constexpr int ITEMS_W = BM * 4 / NW;                 // (row, k-octet) items per wave
        constexpr int NIT = (ITEMS_W + 63) / 64;             // per lane: 1 (BM = 128) or 2 (BM = 160: lanes 0..15 take a second one)
        static_assert(NIT <= RB && NIT <= 2, "the splits ride in the last NIT token blocks of a tile");
        int ivoff[NIT], ildso[NIT];
        bool ion[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int idx = 64 * u + lane;
            ion[u] = idx < ITEMS_W;
            const int t = wave_u * ITEMS_W + (ion[u] ? idx : 0);
            const int r = t >> 2, q = t & 3;                 // LDS row, k-octet
            const int sc = r / (16 * RB), tk = r % (16 * RB), gr = sc * p.ntok + tk;
            ivoff[u] = (tk < p.ntok && gr < rows_here ? gr : 0) * p.lda * 4 + q * 32;
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
            const char* base = (const char*)(seg1 ? xb : xb2) + (seg1 ? k0 : k0 - p.k1) * 4;
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                ld[u][0] = *(const f32x4*)(base + ivoff[u]);
                ld[u][1] = *(const f32x4*)(base + ivoff[u] + 16);
            }
        };
        auto store_item = [&](int u, char* stage) {
            bf16x8 a, b, c;
            split8(ld[u][0], ld[u][1], a, b, c);
            // idle lanes of a ragged round (they hold a copy of item 0) write to a dump slot instead of being masked off:
            // an exec-masked store would cut the MFMA stream into basic blocks
            char* d0 = ion[u] ? stage + ildso[u] : smem + 2 * STAGE + lane * 16;
            const int ps = ion[u] ? X_PLANE : 0;
            *(bf16x8*)d0 = a;
            *(bf16x8*)(d0 + ps) = b;
            *(bf16x8*)(d0 + 2 * ps) = c;
        };
        // prologue: tile 0 -> stage 0 (the weight DMA of tile 0 was issued above)
        load_items(0);
#pragma unroll
        for (int u = 0; u < NIT; ++u) store_item(u, smem);
        // attribution probes (WRONG results, timing only), bits of p.act: 64 no split + plane writes, 128 token fragments read once,
        // 256 no weight DMA, 512 weight fragments read once, 1024 no block barrier, 2048 no token-row loads
        const bool pr_nosplit = p.act & 64, pr_xonce = p.act & 128, pr_nodma = p.act & 256, pr_wonce = p.act & 512,
                   pr_nobar = p.act & 1024, pr_noload = p.act & 2048;
        bf16x8 wf[4][3], xf[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) xf[0][pl] = xf[1][pl] = *(const bf16x8*)(smem + pl * X_PLANE + xoff[0]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wf[j][pl] = *(const bf16x8*)(smem + woff[j] + pl * B_PLANE);
        for (int kt = 0; kt < KT; ++kt) {
            __builtin_amdgcn_s_waitcnt(0x0070);              // my DMA chunks and my plane writes of tile kt are done
            if (!pr_nobar) __syncthreads();                  // everyone's are; nobody reads the other stage any more
            char* cur = smem + (kt & 1) * STAGE;
            char* nxt = smem + ((kt + 1) & 1) * STAGE;
            const int kn = min(kt + 1, KT - 1);              // the tail re-stages the last tile (no branch in the loop)
            if (!pr_nodma) dma_tile(kn, nxt);
            if (!pr_noload) load_items(kn);
#ifdef RES_WARM
            // pull the residual tile (BM rows x BN f32 = 128-byte lines) towards L2 / MALL under the K loop: one dead 4-byte load per
            // line, a few lines per thread, spread over the first K tiles -- the epilogue's residual read then no longer joins the
            // HBM burst of the output stores
            if (p.residual && kt < WARM_TILES) {
                constexpr int LINES = BM * (BN * 4 / 128);                    // lines of the block's residual tile
                constexpr int PER_T = (LINES + T * WARM_TILES - 1) / (T * WARM_TILES);
#pragma unroll
                for (int u = 0; u < PER_T; ++u) {
                    const int ln = (kt * PER_T + u) * T + tid;
                    const int r = ln / (BN * 4 / 128), c = ln % (BN * 4 / 128);
                    const int sc = r / (16 * RB), tk = r % (16 * RB), gr = sc * p.ntok + tk;
                    if (ln < LINES && tk < p.ntok && gr < rows_here) {
                        float dead;
                        asm volatile("global_load_dword %0, %1, off" : "=v"(dead) : "v"(p.residual + (int64_t)(row0 + gr) * p.ldr + col0 + c * 32));
                    }
                }
            }
#endif
            if (!pr_xonce) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) xf[0][pl] = *(const bf16x8*)(cur + pl * X_PLANE + xoff[0]);
            }
            if (!pr_wonce) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) wf[j][pl] = *(const bf16x8*)(cur + woff[j] + pl * B_PLANE);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                if (i + 1 < RB && !pr_xonce) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) xf[(i + 1) & 1][pl] = *(const bf16x8*)(cur + pl * X_PLANE + xoff[i + 1]);
                }
                if (i == RB - NIT) {
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_waitcnt(0x0f70);      // the staged f32 rows (and the weight DMA) of the next tile have arrived
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (i >= RB - NIT && !pr_nosplit) store_item(i - (RB - NIT), nxt);
                mma_block(wf, xf[i & 1][0], xf[i & 1][1], xf[i & 1][2], acc[i]);
                // MFMA first, then the fragment reads of the next block, then the split (2 VALU per MFMA) and its 3 LDS writes
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i + 1 < RB) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
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
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);

    // epilogue: lane = token (block i, row l15) x 4 consecutive channels per MFMA block j; wave = scene wm x group wn
    const int cbase = col0 + wn * 64 + 4 * g;
    const int srow = wm * p.ntok;                        // first row of this wave's scene, relative to row0
    bool valid[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) valid[i] = i * 16 + l15 < p.ntok && srow + i * 16 + l15 < rows_here;
    float* const ob = p.out + (int64_t)(row0 + srow + l15) * p.ldc + cbase;
    const float* const rbp = p.residual ? p.residual + (int64_t)(row0 + srow + l15) * p.ldr + cbase : nullptr;
    if constexpr (!GN) {
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (valid[i]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 y = acc[i][j];
                    if (p.act == 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = 0.5f * y[e] * (1.f + erff(y[e] * 0.70710678118654752f));
                    } else if (p.act == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = y[e] / (1.f + __expf(-y[e]));
                    }
                    if (rbp) y += *(const f32x4*)(rbp + (int64_t)i * 16 * p.ldr + j * 16);
                    *(f32x4*)(ob + (int64_t)i * 16 * p.ldc + j * 16) = y;
                }
            }
        }
    } else {
        // The wave tile IS one GroupNorm cell: the ntok tokens of scene (row0 / ntok + wm) x the 64 channels of group
        // (col0 / 64 + wn).  Statistics are wave-local (two passes over the lane's accumulators + a wave sum; padding rows
        // masked): no LDS, no block barrier.
        const int scene = min(row0 / p.ntok + wm, p.m / p.ntok - 1);
        const bool per_row = p.ss_mode == 1 || p.ss_mode == 3;       // scale/shift differ per token: applied per element below
        f32x4 ga[4], be[4], sc[4], sh[4];
        int64_t ssrow = scene;
        if (p.ss_mode == 4) ssrow = p.ss_index[scene];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                    // issued first: their latency hides under the statistics
            ga[j] = *(const f32x4*)(p.gamma + cbase + j * 16);
            be[j] = *(const f32x4*)(p.beta + cbase + j * 16);
            if (p.ss_mode == 2 || p.ss_mode == 4) {
                const float* ssr = p.scale_shift + ssrow * p.ld_ss + cbase + j * 16;
                sc[j] = *(const f32x4*)ssr;
                sh[j] = *(const f32x4*)(ssr + p.n);
            } else {
                sc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                sh[j] = sc[j];
            }
        }
#ifdef RES_EARLY
        f32x4 res[RB][4];
        if (rbp) {
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    res[i][j] = valid[i] ? *(const f32x4*)(rbp + (int64_t)i * 16 * p.ldr + j * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#endif
        if (p.preact) {
            float* const pb = p.preact + (int64_t)(row0 + srow + l15) * p.ld_pre + cbase;
#pragma unroll
            for (int i = 0; i < RB; ++i)
                if (valid[i]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) *(f32x4*)(pb + (int64_t)i * 16 * p.ld_pre + j * 16) = acc[i][j];
                }
        }
        auto wave_sum = [](float v) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            return v;
        };
        const float inv_cnt = 1.f / (float)(p.ntok * 64);
        float s0 = 0.f;
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                s0 += valid[i] ? (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]) : 0.f;
        const float mean = wave_sum(s0) * inv_cnt;
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
        const float rstd = 1.f / sqrtf(wave_sum(q0) * inv_cnt + p.eps);
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
                        const int64_t row = p.ss_mode == 1 ? (int64_t)row0 + srow + i * 16 + l15 : (int64_t)i * 16 + l15;
                        const float* ssr = p.scale_shift + row * p.ld_ss + cbase + j * 16;
                        s1 = *(const f32x4*)ssr + 1.f;
                        s2 = *(const f32x4*)(ssr + p.n);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = fmaf(acc[i][j][e], ga[j][e], be[j][e]);
                        if (per_row) t = fmaf(t, s1[e], s2[e]);
                        y[e] = t / (1.f + __expf(-t));
                    }
#ifdef RES_EARLY
                    if (rbp) y += res[i][j];
#else
                    if (rbp) y += *(const f32x4*)(rbp + (int64_t)i * 16 * p.ldr + j * 16);
#endif
                    *(f32x4*)(ob + (int64_t)i * 16 * p.ldc + j * 16) = y;
                }
            }
        }
    }
}

}  // namespace

// One launcher for every variant.  gn: fused Block epilogue.  products: 6 = f32-accurate, 3 = "bf16x3" (about 2^-17),
// 1 = plain bf16 (the pipe's ceiling with the same data movement).  pipe: 0 compiler-scheduled, 1 split pipelined inside a K
// tile, 2 across K tiles, 3 split once per block by the staging threads (token tile kept as three bf16 planes in LDS).
template <int WM, int WN, int RB>
static int launch_cfg(const Args& a, int gn, int products, int pipe, hipStream_t s) {
    using C = Cfg<WM, WN, RB>;
    if (a.n % C::BN || a.ntok > 16 * RB || a.ntok <= 0) return 2;
    if ((int64_t)WM * a.ntok * a.lda * 4 >= 0x7fffffffLL || 3LL * a.n * a.k * 2 >= 0x7fffffffLL) return 3;   // 32-bit DMA offsets
    const int scenes = (a.m + a.ntok - 1) / a.ntok;
    const unsigned grid = (unsigned)(((scenes + WM - 1) / WM) * (a.n / C::BN));
    if (gn) {
        if (products != 6 || a.m % a.ntok) return 2;
        if (pipe == 1) gemm_bf16_split_kernel<6, 1, true, WM, WN, RB><<<grid, T, 0, s>>>(a);
        else if (pipe == 2 && a.k % 64 == 0) gemm_bf16_split_kernel<6, 2, true, WM, WN, RB><<<grid, T, 0, s>>>(a);
        else if (pipe == 3) { if constexpr (C::P3) gemm_bf16_split_kernel<6, 3, true, WM, WN, RB><<<grid, T, 0, s>>>(a); else return 2; }
        else return 2;
    } else if (products == 6 && pipe == 0) gemm_bf16_split_kernel<6, 0, false, WM, WN, RB><<<grid, T, 0, s>>>(a);
    else if (products == 6 && pipe == 1) gemm_bf16_split_kernel<6, 1, false, WM, WN, RB><<<grid, T, 0, s>>>(a);
    else if (products == 6 && pipe == 2 && a.k % 64 == 0) gemm_bf16_split_kernel<6, 2, false, WM, WN, RB><<<grid, T, 0, s>>>(a);
    else if (products == 6 && pipe == 3) { if constexpr (C::P3) gemm_bf16_split_kernel<6, 3, false, WM, WN, RB><<<grid, T, 0, s>>>(a); else return 2; }
    else if (products == 1 && pipe == 3) { if constexpr (C::P3) gemm_bf16_split_kernel<1, 3, false, WM, WN, RB><<<grid, T, 0, s>>>(a); else return 2; }
    else if (products == 3 && pipe == 2 && a.k % 64 == 0) gemm_bf16_split_kernel<3, 2, false, WM, WN, RB><<<grid, T, 0, s>>>(a);
    else if (products == 1 && pipe == 0) gemm_bf16_split_kernel<1, 0, false, WM, WN, RB><<<grid, T, 0, s>>>(a);
    else return 2;
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

extern "C" {

int bf16x6_split_planes(const float* w, long count, uint16_t* planes, hipStream_t s) {
    const long pairs = (count + 1) / 2;
    split_planes_kernel<<<(unsigned)((pairs + 255) / 256), 256, 0, s>>>(w, count, planes);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// tile: 0 = automatic, 1 = 2x4 waves (160 x 256 at RB 5), 2 = 4x2 waves (320 x 128 at RB 5).  ntok: tokens per scene for the
// fused epilogue (80 -> RB 5; <= 32, e.g. 21 -> RB 2 with 4 scenes x 128 channels per block); 0 = dense rows (plain GEMM, RB 5).
int bf16x6_launch(const float* x, const float* x2, int k1, int lda, const uint16_t* planes, const float* bias, float* out, int ldc,
                  int m, int n, int k, int act, const float* residual, int ldr, int gn, const float* gamma, const float* beta,
                  float eps, const float* scale_shift, int ld_ss, int ss_mode, const int64_t* ss_index, float* preact, int ld_pre, int ntok,
                  int products, int pipe, int tile, hipStream_t s) {
    if (k % BK || k1 % BK || k1 <= 0 || k1 > k || (k1 < k && !x2) || (lda & 3) || (ldc & 3) || (ldr & 3) || (ld_ss & 3) || m <= 0)
        return 2;
    if (gn && (ntok <= 0 || !gamma || !beta || ss_mode < 0 || ss_mode > 4 || (ss_mode && !scale_shift) || (ss_mode == 4 && !ss_index) ||
               (ld_pre & 3)))
        return 2;
    const bool small = ntok > 0 && ntok <= 32;
    Args a{x, k1 < k ? x2 : nullptr, k1, planes, bias, out, m, n, k, lda, ldc, small ? ntok : (ntok > 0 ? ntok : 80),
           gamma, beta, eps, scale_shift, ld_ss, scale_shift ? ss_mode : 0, ss_index, preact, ld_pre, residual, ldr, act};
    if (small) return launch_cfg<4, 2, 2>(a, gn, products, pipe, s);
    if (tile == 1 || (tile == 0 && n % 256 == 0)) return launch_cfg<2, 4, 5>(a, gn, products, pipe, s);
    return launch_cfg<4, 2, 5>(a, gn, products, pipe, s);
}

}  // extern "C"

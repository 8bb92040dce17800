#pragma once
// fp32 MFMA GEMM with fused epilogues for the Unet1D denoiser (gfx950).
//
// Every 1x1 conv / linear of the reference denoiser (denoise_net.py) is  Y[m][n] = X[m][k] . W[n][k]^T:
// both operands K-contiguous.  The kernel computes the TRANSPOSED product with
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain):  D[channel][token] = sum_k W[channel][k] X[token][k]
//   A operand (32 "rows")  = 32 output channels of W,   lane l supplies W[c0 + (l&31)][k + (l>>5)]
//   B operand (32 "cols")  = 32 tokens of X,            lane l supplies X[t0 + (l&31)][k + (l>>5)]
//   D: lane l holds token t0 + (l&31) and channels c0 + (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15.
// In that orientation every per-token reduction over channels (LayerNorm-like) is in-lane, a lane's
// register quad is 4 CONSECUTIVE channels (16-byte stores), and all tokens of a scene sit in lanes of the
// same block, so GroupNorm (64 channels x N tokens of one scene, denoise_net.py:164) is computed from the
// accumulators: the whole Block.forward (WS-conv + GroupNorm + scale/shift + SiLU, :167-176) is one kernel.
//
// Tiling: 256 threads = 4 waves; block tile BM tokens x 128 channels; BK = 32 staged through LDS with a
// padded row stride of 36 floats (ds_read_b128 of 16 rows is bank-conflict-free: 36*r mod 64 hits 16
// distinct 4-bank slots).  A lane reads 4 consecutive k (one ds_read_b128) per fragment and feeds 4 MFMAs
// (lanes 0-31 carry k..k+3, lanes 32-63 carry k+4..k+7: the k-permutation is the same for both operands).
// Register-staged prefetch of tile kt+1 overlaps the MFMAs of tile kt; two blocks per CU cover barriers.

#include <type_traits>
#include "dsc_common.h"

#ifdef DSC_GEMM_TIMING          // tools/gemm_tune.hip only: per-block phase timestamps (shader clock), start stagger
extern __device__ long long g_dsc_timing[];
extern __device__ int g_dsc_stagger;      // cycles by which the second resident block of every CU starts late
#define DSC_STAMP(i) do { if (threadIdx.x == 0) g_dsc_timing[(blockIdx.x & 4095) * 8 + (i)] = clock64(); } while (0)
#define DSC_STAGGER() do { if (g_dsc_stagger > 0 && blockIdx.x >= gridDim.x / 2) {                               \
        const long long t0__ = clock64(); while (clock64() - t0__ < g_dsc_stagger) __builtin_amdgcn_s_sleep(8); } } while (0)
#else
#define DSC_STAMP(i) do {} while (0)
#define DSC_STAGGER() do {} while (0)
#endif

namespace dsc_gemm {

// TM x TN : 32x32 MFMA tiles per wave;  WM x WN : waves per block (4 or 8);  BK : K elements per staged tile;
// DB : double-buffered LDS (one barrier per K tile);  MINW : min waves per SIMD for __launch_bounds__;
// XCD : remap block ids so the column blocks that share a token tile run on the same XCD (shared L2).
// PIPE (needs DB): MFMA fragments are software-pipelined one 8-wide K step ahead in registers, across the single
// barrier per K tile, so one wave per SIMD can keep the matrix pipe busy on its own.
template <int TM, int TN, int WM, int WN, bool GN, int BK = 32, bool DB = false, int MINW = 2, bool XCD = false,
          bool PIPE = false, bool EPF = false, int PROBE = 0, bool IL = false>
__global__ __launch_bounds__(64 * WM * WN, MINW) void gemm_kernel(const dsc_gemm_args p, const int ncolblk) {
    constexpr int BM = 32 * TM * WM;
    constexpr int BN = 32 * TN * WN;
    constexpr int NW = WM * WN;
    constexpr int T = 64 * NW;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per block");
    constexpr int LDT = BK + 4;               // padded LDS row: ds_read_b128 of 16 consecutive rows hits 16 distinct 4-bank slots
    constexpr int KQ = BK / 4;                // float4 per tile row
    constexpr int XTOT = BM * KQ, WTOT = BN * KQ;
    constexpr int XF = (XTOT + T - 1) / T;
    constexpr int WF = (WTOT + T - 1) / T;
    constexpr int STAGE = (BM + BN) * LDT;
    constexpr bool XFULL = (XTOT % T) == 0, WFULL = (WTOT % T) == 0;

    // epilogue scratch: GroupNorm partials (sum + centred sum of squares) | stats + row tables | per-wave transpose patches
    constexpr int SSL_MAX = 8;
    constexpr int EPI = (GN ? 2 * (BN / 32) * BM + 512 : (BN / 32) * BM + 512) + NW * 32 * 36;
    constexpr int IL_STAGE = (BM + BN) * BK;  // IL: unpadded rows, XOR-swizzled 16-byte slots, two stages
    constexpr int MAINF = IL ? 2 * IL_STAGE : (DB ? 2 : 1) * STAGE;
    constexpr int SMEM = (MAINF > EPI) ? MAINF : EPI;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    // (scale, shift) rows of the block's scenes: written once at kernel start, read by the store loop -- its own array because
    // the staging buffers of the main loop occupy smem in between
    __shared__ __attribute__((aligned(16))) float ssl[GN ? SSL_MAX * 2 * BN : 4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;

    const int z = blockIdx.y;
    int bid = blockIdx.x;
    if (XCD) {
        const int nb = gridDim.x;
        if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);   // XCD x gets the contiguous chunk [x*nb/8, (x+1)*nb/8)
    }
    const int rb = bid / ncolblk, cb = bid % ncolblk;

    const int N = GN ? p.tokens_per_scene : 1;
    const int rows_per_blk = GN ? (BM / N) * N : BM;
    const int64_t row0 = (int64_t)rb * rows_per_blk;
    const int rows_here = (int)((p.m - row0) < (int64_t)rows_per_blk ? (p.m - row0) : (int64_t)rows_per_blk);
    const int col0 = cb * BN;

    const float* a1 = p.a1 + (int64_t)z * p.sa1;
    const float* a2 = p.a2 ? p.a2 + (int64_t)z * p.sa2 : nullptr;
    const float* w = p.w + (int64_t)z * p.sw;
    const float* bias = p.bias ? p.bias + (int64_t)z * p.sbias : nullptr;
    const float* res = p.residual ? p.residual + (int64_t)z * p.sres : nullptr;
    float* y = p.y + (int64_t)z * p.sy;

    const int nk = (p.k1 + p.k2) / BK;

    f32x4 xr[XF], wr[WF];
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.0f;

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        const float* ab;
        int64_t lda;
        int kk;
        if (k0 < p.k1) { ab = a1; lda = p.lda1; kk = k0; }
        else           { ab = a2; lda = p.lda2; kk = k0 - p.k1; }
#pragma unroll
        for (int i = 0; i < XF; ++i) {
            // branch-free: out-of-range rows read row 0 of the tile (always valid); they are zeroed when staged, so the
            // wait for the load sits at the ds_write one tile later, not here.  IL: threads past the end of a ragged last
            // round wrap to the start of the tile (a duplicate of what another thread stages -- same address, same data)
            int f = tid + T * i;
            if (IL && !XFULL && f >= XTOT) f -= XTOT;
            const int r = f / KQ, kq = f % KQ;
            const bool ok = (IL || XFULL || f < XTOT) && r < rows_here;
            xr[i] = *reinterpret_cast<const f32x4*>(ab + (row0 + (ok ? r : 0)) * lda + kk + kq * 4);
        }
#pragma unroll
        for (int i = 0; i < WF; ++i) {
            int f = tid + T * i;
            if (IL && !WFULL && f >= WTOT) f -= WTOT;
            const int r = f / KQ, kq = f % KQ;
            const int c = col0 + r;
            const bool ok = (IL || WFULL || f < WTOT) && c < p.n;
            wr[i] = *reinterpret_cast<const f32x4*>(w + (int64_t)(ok ? c : col0) * p.ldw + k0 + kq * 4);
        }
    };
    auto store_tile = [&](float* stage) {
        float* Xs = stage;
        float* Ws = stage + BM * LDT;
#pragma unroll
        for (int i = 0; i < XF; ++i) {
            const int f = tid + T * i;
            const int r = f / KQ, kq = f % KQ;
            const f32x4 v = (r < rows_here) ? xr[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            if (XFULL || f < XTOT) *reinterpret_cast<f32x4*>(Xs + r * LDT + kq * 4) = v;
        }
#pragma unroll
        for (int i = 0; i < WF; ++i) {
            const int f = tid + T * i;
            const int r = f / KQ, kq = f % KQ;
            const f32x4 v = (col0 + r < p.n) ? wr[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            if (WFULL || f < WTOT) *reinterpret_cast<f32x4*>(Ws + r * LDT + kq * 4) = v;
        }
    };
    auto compute_tile = [&](const float* stage) {
        const float* Xs = stage;
        const float* Ws = stage + BM * LDT;
#pragma unroll
        for (int k8 = 0; k8 < BK / 8; ++k8) {
            f32x4 xf[TM], wf[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                xf[tm] = *reinterpret_cast<const f32x4*>(Xs + ((wm * TM + tm) * 32 + l31) * LDT + k8 * 8 + half * 4);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                wf[tn] = *reinterpret_cast<const f32x4*>(Ws + ((wn * TN + tn) * 32 + l31) * LDT + k8 * 8 + half * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[tn][j], xf[tm][j], acc[tm][tn], 0, 0, 0);
        }
    };

#ifdef DSC_GEMM_TIMING
#endif
    DSC_STAGGER();
    DSC_STAMP(0);
    // GroupNorm epilogue inputs that do not depend on the product are fetched BEFORE the main loop: (scale, shift) of a
    // time-conditioned block is one row per scene; the block's rows go to LDS here (their latency overlaps the first operand
    // tile's) instead of being gathered row by row between the output stores -- the compiler cannot hoist those gathers over
    // the stores (possible aliasing), which serialised one L2 round trip per 8 output rows -- and nothing in the epilogue has
    // to wait behind the residual prefetch for them (s_waitcnt vmcnt retires loads in order).
    constexpr int SSV = GN ? (SSL_MAX * 2 * BN / T) : 1;
    const int spt = GN ? BM / N : 1;
    const int scenes_here = GN ? rows_here / N : 0;
    const int64_t scene0 = (int64_t)rb * spt;                 // blocks are scene-aligned
    const bool has_ss = GN && p.scale_shift != nullptr;
    const bool ss_lds = has_ss && (p.ss_mode == DSC_SS_PER_SCENE || p.ss_mode == DSC_SS_BY_INDEX) && scenes_here <= SSL_MAX;
    load_tile(0);
    if constexpr (GN) {
        if (ss_lds) {
#pragma unroll
            for (int j = 0; j < SSV; ++j) {
                const int f = tid + T * j;                     // (scene, half, column) flattened
                const int sc = f / (2 * BN), hc = f % (2 * BN);
                if (sc < scenes_here) {
                    const int64_t row = (p.ss_mode == DSC_SS_PER_SCENE) ? scene0 + sc : p.ss_index[scene0 + sc];
                    ssl[f] = p.scale_shift[row * p.ld_ss + (hc >= BN ? p.n : 0) + col0 + (hc % BN)];
                }
            }
        }
    }
    // Outputs leave through LDS (see the epilogue); the flags are needed before the main loop because the residual prefetch
    // is issued inside its last iterations.
    const bool rfast = res && ((p.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(res) & 15) == 0);
    const bool fast = ((p.ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0) && (col0 + BN <= p.n) &&
                      (!res || rfast);
    const int tr = lane >> 3, cq = lane & 7;
    // EPF: every residual quad this lane will add is requested at the top of the epilogue, ahead of the GroupNorm statistics,
    // so the HBM burst of the residual stream (42 MB per layer at M = 20480) is not paid tile by tile between the stores.
    f32x4 rpre[EPF ? TM * TN * 4 : 1];
    const bool use_pre = EPF && rfast && (GN || fast);
    auto prefetch_residual = [&]() {
        if constexpr (EPF) {
            if (use_pre) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int tl = (wm * TM + tm) * 32 + tr + 8 * i;
                            const int c = col0 + (wn * TN + tn) * 32 + cq * 4;
                            const int tlc = tl < rows_here ? tl : 0;
                            rpre[(tn * TM + tm) * 4 + i] = *reinterpret_cast<const f32x4*>(res + (row0 + tlc) * p.ldr + c);
                        }
            }
        }
    };
    if constexpr (IL) {
        // Interleaved main loop.  One K tile = S groups of TM*TN*4 MFMAs; everything else a K tile needs -- the next group's
        // fragment reads, the LDS writes of tile kt+1 (register-staged one tile ahead), the global loads of tile kt+2 -- is
        // slotted BETWEEN the MFMAs of this wave's own instruction stream (sched_group_barrier), because the two waves of a SIMD
        // belong to the same block and run in lock-step: a stretch of non-MFMA instructions leaves the matrix pipe idle
        // (measured: 12 k of 178 k main-loop cycles at K = 512 for the grouped order).  LDS rows are unpadded (BK floats); the
        // 16-byte slot of k-quad q in row r sits at q ^ ((r >> 1) & 7): fragment reads (16 consecutive rows, one quad) and
        // staging writes (2 rows x 8 quads) both touch 16 distinct slots -> conflict-free without padding.
        static_assert(BK == 32, "IL: BK = 32 (8 quads per row)");
        constexpr int S = BK / 8;
        constexpr int MM = TM * TN * 4;           // MFMAs per group
        constexpr int NLD = XF + WF;              // staging instructions (global loads / LDS writes) per thread and tile
        static_assert(MM >= TM + TN + NLD, "IL: the MFMA groups are too short to carry the staging instructions");
        const int swz = (l31 >> 1) & 7;
        int foff[S];                              // float offset of this lane's fragment quad for each 8-wide K step
#pragma unroll
        for (int k8 = 0; k8 < S; ++k8) foff[k8] = (((k8 * 2 + half) ^ swz) << 2);
        f32x4 xfA[TM], wfA[TN], xfB[TM], wfB[TN];
        auto frags = [&](const float* stage, int k8, f32x4 (&xf)[TM], f32x4 (&wf)[TN]) {
            const float* Xs = stage;
            const float* Ws = stage + BM * BK;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                xf[tm] = *reinterpret_cast<const f32x4*>(Xs + ((wm * TM + tm) * 32 + l31) * BK + foff[k8]);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                wf[tn] = *reinterpret_cast<const f32x4*>(Ws + ((wn * TN + tn) * 32 + l31) * BK + foff[k8]);
        };
        auto mma = [&](const f32x4 (&xf)[TM], const f32x4 (&wf)[TN]) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[tn][j], xf[tm][j], acc[tm][tn], 0, 0, 0);
        };
        // rows / columns past the edge were loaded from a valid row and are NOT zeroed: they only feed accumulators of
        // rows / columns that are never stored (nor enter the GroupNorm statistics)
        auto stage_tile = [&](float* stage) {
            float* Xs = stage;
            float* Ws = stage + BM * BK;
#pragma unroll
            for (int i = 0; i < XF; ++i) {
                int f = tid + T * i;
                if (!XFULL && f >= XTOT) f -= XTOT;          // duplicate slot, identical data (see load_tile): no branch
                const int r = f / KQ, kq = f % KQ;
                *reinterpret_cast<f32x4*>(Xs + r * BK + ((kq ^ ((r >> 1) & 7)) << 2)) = xr[i];
            }
#pragma unroll
            for (int i = 0; i < WF; ++i) {
                int f = tid + T * i;
                if (!WFULL && f >= WTOT) f -= WTOT;
                const int r = f / KQ, kq = f % KQ;
                *reinterpret_cast<f32x4*>(Ws + r * BK + ((kq ^ ((r >> 1) & 7)) << 2)) = wr[i];
            }
        };
        stage_tile(smem);
        load_tile(nk > 1 ? 1 : 0);
        __syncthreads();
        DSC_STAMP(1);
        frags(smem, 0, xfA, wfA);
        for (int kt = 0; kt < nk; ++kt) {
            const float* cur = smem + (kt & 1) * IL_STAGE;
            float* nxt = smem + ((kt + 1) & 1) * IL_STAGE;
            __builtin_amdgcn_sched_barrier(0);
            // group 0: MFMAs(A) with the fragment reads of group 1
            frags(cur, 1, xfB, wfB);
            mma(xfA, wfA);
#pragma unroll
            for (int i = 0; i < TM + TN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, MM - 2 * (TM + TN), 0);
            __builtin_amdgcn_sched_barrier(0);
            // group 1: MFMAs(B) with the fragment reads of group 2 and the LDS writes of tile kt+1
            frags(cur, 2, xfA, wfA);
            stage_tile(nxt);
            mma(xfB, wfB);
#pragma unroll
            for (int i = 0; i < TM + TN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, MM - (TM + TN) - NLD, 0);
            __builtin_amdgcn_sched_barrier(0);
            // group 2: MFMAs(A) with the fragment reads of group 3 and the global loads of tile kt+2
            frags(cur, 3, xfB, wfB);
            load_tile(kt + 2 < nk ? kt + 2 : nk - 1);
            mma(xfA, wfA);
#pragma unroll
            for (int i = 0; i < TM + TN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, MM - (TM + TN) - NLD, 0);
            __builtin_amdgcn_sched_barrier(0);
            // group 3: MFMAs(B); the block barrier (tile kt+1 complete in LDS, tile kt's reads retired) and the first
            // fragment reads of tile kt+1 sit in the middle of the group
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(wfB[tn][j], xfB[tm][j], acc[tm][tn], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            frags(nxt, 0, xfA, wfA);
#pragma unroll
            for (int j = 2; j < 4; ++j)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(wfB[tn][j], xfB[tm][j], acc[tm][tn], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM + TN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, MM / 2 - (TM + TN), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    } else if constexpr (PIPE) {
        static_assert(DB, "PIPE needs the double-buffered LDS stages");
        constexpr int S = BK / 8;                 // 8-wide K steps per staged tile
        static_assert(S == 2 || S == 4, "BK must be 16 or 32");
        f32x4 xfA[TM], wfA[TN], xfB[TM], wfB[TN];
        auto frags = [&](const float* stage, int k8, f32x4 (&xf)[TM], f32x4 (&wf)[TN]) {
            const float* Xs = stage;
            const float* Ws = stage + BM * LDT;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                xf[tm] = *reinterpret_cast<const f32x4*>(Xs + ((wm * TM + tm) * 32 + l31) * LDT + k8 * 8 + half * 4);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                wf[tn] = *reinterpret_cast<const f32x4*>(Ws + ((wn * TN + tn) * 32 + l31) * LDT + k8 * 8 + half * 4);
        };
        auto mma = [&](const f32x4 (&xf)[TM], const f32x4 (&wf)[TN]) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[tn][j], xf[tm][j], acc[tm][tn], 0, 0, 0);
        };
        store_tile(smem);
        __syncthreads();
        DSC_STAMP(1);
        // branch-free, schedule-pinned pipeline: the compiler otherwise sinks the prefetches next to their uses and waits on
        // them immediately (seen in the ISA); clamped tile indices make the last iterations re-stage the final tile
        load_tile(nk > 1 ? 1 : 0);
        frags(smem, 0, xfA, wfA);
        for (int kt = 0; kt < nk; ++kt) {
            const float* cur = smem + (kt & 1) * STAGE;
            float* nxt = smem + ((kt + 1) & 1) * STAGE;
            if constexpr (S == 4) {
                frags(cur, 1, xfB, wfB);
                __builtin_amdgcn_sched_barrier(0);
                mma(xfA, wfA);
                __builtin_amdgcn_sched_barrier(0);
                frags(cur, 2, xfA, wfA);
                __builtin_amdgcn_sched_barrier(0);
                mma(xfB, wfB);
                __builtin_amdgcn_sched_barrier(0);
                frags(cur, 3, xfB, wfB);
                store_tile(nxt);
                __builtin_amdgcn_sched_barrier(0);
                mma(xfA, wfA);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                load_tile(kt + 2 < nk ? kt + 2 : nk - 1);
                frags(nxt, 0, xfA, wfA);
                __builtin_amdgcn_sched_barrier(0);
                mma(xfB, wfB);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                frags(cur, 1, xfB, wfB);
                store_tile(nxt);
                __builtin_amdgcn_sched_barrier(0);
                mma(xfA, wfA);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                load_tile(kt + 2 < nk ? kt + 2 : nk - 1);
                frags(nxt, 0, xfA, wfA);
                __builtin_amdgcn_sched_barrier(0);
                mma(xfB, wfB);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    } else if constexpr (DB) {
        // one barrier per K tile: tile kt+1 is written into the other LDS stage while tile kt is being consumed
        store_tile(smem);
        __syncthreads();
        DSC_STAMP(1);
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_tile(kt + 1);
            compute_tile(smem + (kt & 1) * STAGE);
            if (kt + 1 < nk) store_tile(smem + ((kt + 1) & 1) * STAGE);
            __syncthreads();
        }
    } else if constexpr (PROBE > 0) {
        // tools/gemm_tune.hip only -- attribution probes of the main loop (results are wrong on purpose):
        //   1: no global loads / LDS stores after the first tile   2: + fragments read once   3: + no barriers
        store_tile(smem);
        __syncthreads();
        DSC_STAMP(1);
        f32x4 xf0[TM], wf0[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) xf0[tm] = *reinterpret_cast<const f32x4*>(smem + ((wm * TM + tm) * 32 + l31) * LDT + half * 4);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) wf0[tn] = *reinterpret_cast<const f32x4*>(smem + BM * LDT + ((wn * TN + tn) * 32 + l31) * LDT + half * 4);
        for (int kt = 0; kt < nk; ++kt) {
            if constexpr (PROBE == 1) {
                compute_tile(smem);
            } else {
#pragma unroll
                for (int k8 = 0; k8 < BK / 8; ++k8)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn)
                                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf0[tn][j], xf0[tm][j], acc[tm][tn], 0, 0, 0);
            }
            if constexpr (PROBE < 3) { __syncthreads(); __syncthreads(); }
            else __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            store_tile(smem);
            __syncthreads();
            if (kt == 0) DSC_STAMP(1);
            if (kt + 1 < nk) load_tile(kt + 1);
            compute_tile(smem);
            __syncthreads();
        }
    }

    DSC_STAMP(2);
    // ------------------------------------------------------------------ epilogue
    // bias
    if (bias) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = col0 + (wn * TN + tn) * 32 + 8 * q + 4 * half;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float b = (c + e < p.n) ? bias[c + e] : 0.0f;
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) acc[tm][tn][4 * q + e] += b;
                }
            }
    }

    // Outputs leave through LDS: the MFMA layout (lane = token, 16 scattered channels) would issue 64 scattered 16-byte
    // accesses per instruction; each wave transposes its 32x32 tile in a private LDS patch and then touches HBM as
    // 8 token rows x 128 contiguous bytes per instruction (stores, residual, scale/shift, pre-norm copy all coalesced).
    constexpr int TLD = 36;
    constexpr int SCR = GN ? 2 * (BN / 32) * BM + 512 : (BN / 32) * BM + 512;   // scratch below the patches
    static_assert(SCR + NW * 32 * TLD <= SMEM, "epilogue scratch must fit in the LDS allocation");
    float* patch = smem + SCR + wave * (32 * TLD);
    // the residual burst is requested here, at the top of the epilogue, and lands under the statistics.  (Measured: requesting it
    // under the last one or two K tiles slows the main loop by what it saves -- the burst competes with the operand loads.)
    prefetch_residual();

    if constexpr (GN) {
        constexpr int G = BN / 64;        // GroupNorm groups covered by this block
        constexpr int CT = BN / 32;       // 32-channel tiles in the block
        float* P = smem;                  // [CT][BM] per-token sums over the 32 channels of a tile
        float* Q = smem + CT * BM;        // [CT][BM] per-token sums of squares about the token's own mean
        float* stat = smem + 2 * CT * BM; // [spt*G] mean, then [spt*G] rstd  (spt * G <= 80 for N >= 4)
        const int nstat = scenes_here * G;
        const float inv_cnt = 1.0f / (64.0f * (float)N);
        // One pass over the accumulators (Chan's pairwise update): every lane reduces its 16 channels to (sum, centred sum
        // of squares), the two lane halves are merged, then one wave per (scene, group) merges the 2N token entries about
        // their common mean.  No E[x^2] - E[x]^2 cancellation anywhere: same quality as mean-then-variance, half the passes.
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int tl = (wm * TM + tm) * 32 + l31;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                float s16 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) s16 += acc[tm][tn][r];
                const float m16 = s16 * (1.0f / 16.0f);
                float q16 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float d = acc[tm][tn][r] - m16; q16 += d * d; }
                const float so = __shfl_xor(s16, 32, 64), qo = __shfl_xor(q16, 32, 64);
                const float dm = (so - s16) * (1.0f / 16.0f);
                if (half == 0) {
                    P[(wn * TN + tn) * BM + tl] = s16 + so;
                    Q[(wn * TN + tn) * BM + tl] = q16 + qo + dm * dm * 8.0f;     // n_a n_b / (n_a + n_b) = 8
                }
            }
        }
        DSC_STAMP(5);
        __syncthreads();
        DSC_STAMP(6);
        for (int st = wave; st < nstat; st += NW) {
            const int sc = st / G, g = st % G;
            float s = 0.f;
            for (int j = lane; j < 2 * N; j += 64) {
                const int ct = 2 * g + (j >= N ? 1 : 0);
                const int tk = (j >= N ? j - N : j);
                s += P[ct * BM + sc * N + tk];
            }
            const float mu = wave_sum(s) * inv_cnt;
            float q = 0.f;
            for (int j = lane; j < 2 * N; j += 64) {
                const int ct = 2 * g + (j >= N ? 1 : 0);
                const int tk = (j >= N ? j - N : j);
                const float d = P[ct * BM + sc * N + tk] * (1.0f / 32.0f) - mu;
                q += Q[ct * BM + sc * N + tk] + 32.0f * d * d;
            }
            q = wave_sum(q);
            if (lane == 0) {
                stat[st] = mu;
                stat[spt * G + st] = 1.0f / sqrtf(q * inv_cnt + p.eps);
            }
        }
        __syncthreads();
        DSC_STAMP(7);
        // Per-row tables so that the store loop below has no integer division and no conditioning-mode branches:
        // (mean, rstd) of the row's scene for each group (reusing the partial-sum area) and the scale/shift row index
        // (LDS-staged rows: the scene's slot in ssl; otherwise the row of the global table).
        float* rowst = smem;                                        // [G][BM][2]
        int* rowss = reinterpret_cast<int*>(stat + 192);            // [BM]; stat holds at most 2 * 40 * 2 floats
        {
            for (int t = tid; t < BM; t += T) {
                const int sc = t / N;
                const bool ok = t < rows_here;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    rowst[(g * BM + t) * 2 + 0] = ok ? stat[sc * G + g] : 0.f;
                    rowst[(g * BM + t) * 2 + 1] = ok ? stat[spt * G + sc * G + g] : 0.f;
                }
                int ssr = 0;
                if (has_ss && ok) {
                    if (ss_lds) ssr = sc;
                    else if (p.ss_mode == DSC_SS_PER_SCENE) ssr = (int)(scene0 + sc);
                    else if (p.ss_mode == DSC_SS_PER_SLOT) ssr = t - sc * N;
                    else if (p.ss_mode == DSC_SS_BY_INDEX) ssr = (int)p.ss_index[scene0 + sc];
                    else ssr = (int)(row0 + t);
                }
                rowss[t] = ssr;
            }
        }
        __syncthreads();
        DSC_STAMP(3);
        float* zp = p.preact ? p.preact + (int64_t)z * p.sy : nullptr;
        const bool yfast = (p.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
        // normalise, affine, scale/shift, SiLU, residual, store -- in the transposed (row-major) patch layout
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int g = (wn * TN + tn) >> 1;
            const int c = col0 + (wn * TN + tn) * 32 + cq * 4;
            const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
            const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c);
            const int tl0 = wm * TM * 32 + tr;
            float* yp = y + (row0 + tl0) * p.ldy + c;               // running row pointers: +8 rows per step
            float* zq = zp ? zp + (row0 + tl0) * p.ld_preact + c : nullptr;
            const int64_t ystep = 8 * p.ldy, zstep = 8 * p.ld_preact;
            const float* ssb = has_ss ? p.scale_shift + c : nullptr;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[tm][tn][4 * q + e];
                    *reinterpret_cast<f32x4*>(patch + l31 * TLD + 8 * q + 4 * half) = v;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int tl = tl0 + tm * 32 + 8 * i;
                    if (tl < rows_here) {
                        const float mu = rowst[(g * BM + tl) * 2 + 0];
                        const float rs = rowst[(g * BM + tl) * 2 + 1];
                        f32x4 v = *reinterpret_cast<const f32x4*>(patch + (tr + 8 * i) * TLD + cq * 4);
                        if (zq) *reinterpret_cast<f32x4*>(zq) = v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (v[e] - mu) * rs * ga[e] + be[e];
                        if (ss_lds) {
                            const float* ss = ssl + rowss[tl] * (2 * BN) + (wn * TN + tn) * 32 + cq * 4;
                            const f32x4 sc4 = *reinterpret_cast<const f32x4*>(ss);
                            const f32x4 sh4 = *reinterpret_cast<const f32x4*>(ss + BN);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] * (sc4[e] + 1.0f) + sh4[e];
                        } else if (has_ss) {
                            const float* ss = ssb + (int64_t)rowss[tl] * p.ld_ss;
                            const f32x4 sc4 = *reinterpret_cast<const f32x4*>(ss);
                            const f32x4 sh4 = *reinterpret_cast<const f32x4*>(ss + p.n);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] * (sc4[e] + 1.0f) + sh4[e];
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = dsc_silu_fast(v[e]);
                        if (res) {
                            if (EPF && use_pre) {
                                const f32x4 r4 = rpre[(tn * TM + tm) * 4 + i];
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += r4[e];
                            } else if (rfast) {
                                const f32x4 r4 = *reinterpret_cast<const f32x4*>(res + (row0 + tl) * p.ldr + c);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += r4[e];
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += res[(row0 + tl) * p.ldr + c + e];
                            }
                        }
                        if (yfast)
                            *reinterpret_cast<f32x4*>(yp) = v;
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) yp[e] = v[e];
                        }
                    }
                    yp += ystep;
                    if (zq) zq += zstep;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    } else if (fast) {
        // one copy of the store loop per output activation: the erf-GELU polynomial must not sit (branched over) in the
        // plain store path
        auto plain_store = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int c = col0 + (wn * TN + tn) * 32 + cq * 4;
            const int tl0 = wm * TM * 32 + tr;
            float* yp = y + (row0 + tl0) * p.ldy + c;               // running row pointers: +8 rows per step
            const float* rp = res ? res + (row0 + tl0) * p.ldr + c : nullptr;
            const int64_t ystep = 8 * p.ldy, rstep = 8 * p.ldr;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = dsc_act(acc[tm][tn][4 * q + e], ACT);
                    *reinterpret_cast<f32x4*>(patch + l31 * TLD + 8 * q + 4 * half) = v;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int tl = tl0 + tm * 32 + 8 * i;
                    if (tl < rows_here) {
                        f32x4 v = *reinterpret_cast<const f32x4*>(patch + (tr + 8 * i) * TLD + cq * 4);
                        if (res) {
                            const f32x4 r4 = (EPF && use_pre) ? rpre[(tn * TM + tm) * 4 + i]
                                                              : *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += r4[e];
                        }
                        *reinterpret_cast<f32x4*>(yp) = v;
                    }
                    yp += ystep;
                    if (rp) rp += rstep;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        };
        if (p.act_out == DSC_ACT_GELU) plain_store(std::integral_constant<int, DSC_ACT_GELU>{});
        else if (p.act_out == DSC_ACT_SILU) plain_store(std::integral_constant<int, DSC_ACT_SILU>{});
        else plain_store(std::integral_constant<int, DSC_ACT_NONE>{});
    } else {
        // ragged / unaligned outputs (narrow decoder heads written at a column offset of the (M, C) tensor)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int tl = (wm * TM + tm) * 32 + l31;
            if (tl >= rows_here) continue;
            const int64_t tok = row0 + tl;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = col0 + (wn * TN + tn) * 32 + 8 * q + 4 * half;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e < p.n) {
                            float o = dsc_act(acc[tm][tn][4 * q + e], p.act_out);
                            if (res) o += res[tok * p.ldr + c + e];
                            y[tok * p.ldy + c + e] = o;
                        }
                }
            }
        }
    }
    DSC_STAMP(4);
}

}  // namespace dsc_gemm

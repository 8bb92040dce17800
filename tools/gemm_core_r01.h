#pragma once
// ROUND-1 snapshot of diffuscene_amd/csrc/gemm_core.h (git a07da01) under its own namespace: the A/B reference of tools/gemm_tune.py.
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
#include "../diffuscene_amd/csrc/dsc_common.h"

#ifndef DSC_STAMP
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
#endif

namespace dsc_gemm_r01 {

// TM x TN : 32x32 MFMA tiles per wave;  WM x WN : waves per block (4 or 8);  BK : K elements per staged tile;
// DB : double-buffered LDS (one barrier per K tile);  MINW : min waves per SIMD for __launch_bounds__;
// XCD : remap block ids so the column blocks that share a token tile run on the same XCD (shared L2).
// PIPE (needs DB): MFMA fragments are software-pipelined one 8-wide K step ahead in registers, across the single
// barrier per K tile, so one wave per SIMD can keep the matrix pipe busy on its own.
template <int TM, int TN, int WM, int WN, bool GN, int BK = 32, bool DB = false, int MINW = 2, bool XCD = false,
          bool PIPE = false, bool EPF = false>
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

    constexpr int EPI = (BN / 32) * BM + 512 + NW * 32 * 36;      // epilogue scratch: GroupNorm partials + per-wave patches
    constexpr int SMEM = ((DB ? 2 : 1) * STAGE > EPI) ? (DB ? 2 : 1) * STAGE : EPI;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];

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
            // wait for the load sits at the ds_write one tile later, not here
            const int f = tid + T * i;
            const int r = f / KQ, kq = f % KQ;
            const bool ok = (XFULL || f < XTOT) && r < rows_here;
            xr[i] = *reinterpret_cast<const f32x4*>(ab + (row0 + (ok ? r : 0)) * lda + kk + kq * 4);
        }
#pragma unroll
        for (int i = 0; i < WF; ++i) {
            const int f = tid + T * i;
            const int r = f / KQ, kq = f % KQ;
            const int c = col0 + r;
            const bool ok = (WFULL || f < WTOT) && c < p.n;
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
    if (threadIdx.x == 0) {
        g_dsc_timing[(blockIdx.x & 4095) * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        g_dsc_timing[(blockIdx.x & 4095) * 8 + 6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    }
#endif
    DSC_STAGGER();
    DSC_STAMP(0);
    load_tile(0);
    if constexpr (PIPE) {
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
    const bool rfast = res && ((p.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(res) & 15) == 0);
    const bool fast = ((p.ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0) && (col0 + BN <= p.n) &&
                      (!res || rfast);
    constexpr int TLD = 36;
    constexpr int SCR = (BN / 32) * BM + 512;                // GroupNorm scratch (P + stats, N >= 4) lives below the patches
    static_assert(SCR + NW * 32 * TLD <= SMEM, "epilogue scratch must fit in the LDS allocation");
    float* patch = smem + SCR + wave * (32 * TLD);
    const int tr = lane >> 3, cq = lane & 7;

    // EPF: every residual quad this lane will add is requested NOW, ahead of the GroupNorm statistics, so the HBM latency
    // of the residual stream hides behind the stats phase instead of being paid tile by tile (the staging / fragment
    // registers of the main loop are dead here, so the 16*TM*TN extra VGPRs stay inside the main loop's allocation).
    f32x4 rpre[EPF ? TM * TN * 4 : 1];
    const bool use_pre = EPF && rfast && (GN || fast);
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

    if constexpr (GN) {
        constexpr int G = BN / 64;        // GroupNorm groups covered by this block
        constexpr int CT = BN / 32;       // 32-channel tiles in the block
        float* P = smem;                  // [CT][BM] per-token partial sums
        float* stat = smem + CT * BM;     // [spt*G] mean, then [spt*G] rstd
        const int spt = BM / N;
        const int scenes_here = rows_here / N;
        const int nstat = scenes_here * G;
        const float inv_cnt = 1.0f / (64.0f * (float)N);
        int scn[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int tl = (wm * TM + tm) * 32 + l31;
            scn[tm] = tl / N;
        }
        // pass 1: mean
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int tl = (wm * TM + tm) * 32 + l31;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[tm][tn][r];
                s += __shfl_xor(s, 32, 64);
                if (half == 0) P[(wn * TN + tn) * BM + tl] = s;
            }
        }
        __syncthreads();
        for (int st = wave; st < nstat; st += NW) {
            const int sc = st / G, g = st % G;
            float s = 0.f;
            for (int j = lane; j < 2 * N; j += 64) {
                const int ct = 2 * g + (j >= N ? 1 : 0);
                const int tk = (j >= N ? j - N : j);
                s += P[ct * BM + sc * N + tk];
            }
            s = wave_sum(s);
            if (lane == 0) stat[st] = s * inv_cnt;
        }
        __syncthreads();
        // pass 2: variance about the mean
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int tl = (wm * TM + tm) * 32 + l31;
            const bool ok = scn[tm] < scenes_here;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int g = (wn * TN + tn) >> 1;
                const float mu = ok ? stat[scn[tm] * G + g] : 0.f;
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float d = acc[tm][tn][r] - mu; s += d * d; }
                s += __shfl_xor(s, 32, 64);
                if (half == 0) P[(wn * TN + tn) * BM + tl] = s;
            }
        }
        __syncthreads();
        for (int st = wave; st < nstat; st += NW) {
            const int sc = st / G, g = st % G;
            float s = 0.f;
            for (int j = lane; j < 2 * N; j += 64) {
                const int ct = 2 * g + (j >= N ? 1 : 0);
                const int tk = (j >= N ? j - N : j);
                s += P[ct * BM + sc * N + tk];
            }
            s = wave_sum(s);
            if (lane == 0) stat[spt * G + st] = 1.0f / sqrtf(s * inv_cnt + p.eps);
        }
        __syncthreads();
        // Per-row tables so that the store loop below has no integer division and no conditioning-mode branches:
        // (mean, rstd) of the row's scene for each group (reusing the partial-sum area) and the scale/shift row index.
        // Blocks are scene-aligned, so the first scene of the block is rb * spt.
        float* rowst = smem;                                        // [G][BM][2]
        int* rowss = reinterpret_cast<int*>(stat + 192);            // [BM]; stat holds at most 2 * 40 * 2 floats
        {
            const int64_t scene0 = (int64_t)rb * spt;
            for (int t = tid; t < BM; t += T) {
                const int sc = t / N;
                const bool ok = t < rows_here;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    rowst[(g * BM + t) * 2 + 0] = ok ? stat[sc * G + g] : 0.f;
                    rowst[(g * BM + t) * 2 + 1] = ok ? stat[spt * G + sc * G + g] : 0.f;
                }
                int ssr = 0;
                if (p.scale_shift && ok) {
                    if (p.ss_mode == DSC_SS_PER_SCENE) ssr = (int)(scene0 + sc);
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
        const bool has_ss = p.scale_shift != nullptr;
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
                        if (has_ss) {
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

}  // namespace dsc_gemm_r01

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
// Tiling: 4 or 8 waves, each TM x TN tiles of 32 x 32; BK = 32 K elements per LDS stage; a lane reads 4 consecutive k (one
// ds_read_b128) per fragment and feeds 4 MFMAs (lanes 0-31 carry k..k+3, lanes 32-63 carry k+4..k+7: the k-permutation is the
// same for both operands).  Main loop (IL = 2): two LDS stages filled by buffer_load ... lds straight from global memory,
// fragments double-buffered in registers, every non-MFMA instruction of a K tile slotted between its MFMAs, one block barrier
// per K tile -- see the comments at the loop and DESIGN.md section 4.

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

// TM x TN : 32x32 MFMA tiles per wave;  WM x WN : waves per block (4 or 8);  BK = 32 K elements per staged tile.
// IL : main loop -- 2 = interleaved, operands DMA'd straight into swizzled LDS stages (the product form), 1 = interleaved,
//      register-staged (operands with two row strides / offsets beyond 32 bits), 0 = classic single-buffered loop (64 x 64 tiles
//      whose MFMA groups are too short to carry the staging instructions).
// PROBE : tools/gemm_tune.hip only -- attribution probes of the main loop (wrong results on purpose).
// Always on (round-1 measurements): block ids remapped so the column blocks sharing a token tile run on one XCD (shared L2);
// the residual quads of the epilogue requested at its top; two waves per SIMD.
// LNF : the block spans ALL n output channels (gridDim = row blocks only) and the epilogue applies the channel LayerNorm of
//       denoise_net.py:93-102 to every token row -- y = (acc - mean) * rsqrt(var + eps) * gain (+ residual): the out-projection
//       of LinearAttention followed by its LayerNorm and the residual of the PreNorm block in one launch.
template <int TM, int TN, int WM, int WN, bool GN, int IL = 0, int PROBE = 0, bool LNF = false>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_kernel(const dsc_gemm_args p, const int ncolblk) {
    constexpr int BK = 32;
    constexpr bool EPF = !LNF;                 // LNF: 96 accumulator + 96 prefetch registers would not fit next to the LayerNorm state
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
    constexpr int SSL_MAX = (IL && NW == 4 && BM + BN > 256) ? 4 : 8;   // 2 blocks per CU: 2 x (stages + tables) <= 160 KB
    constexpr int EPI = (GN ? 2 * (BN / 32) * BM + 512 : (BN / 32) * BM + 512) + NW * 32 * 36;
    constexpr int IL_STAGE = (BM + BN) * BK;  // IL: unpadded rows, XOR-swizzled 16-byte slots, two stages
    constexpr int MAINF = IL ? 2 * IL_STAGE : STAGE;
    constexpr int SMEM = (MAINF > EPI) ? MAINF : EPI;
    // ONE __shared__ object (a second one makes hipcc drain the LDS-DMA queue before every fragment read):
    // [ operand stages / epilogue scratch | (scale, shift) rows of the block's scenes | per-row tables of the GN epilogue ]
    constexpr int SSLF = GN ? SSL_MAX * 2 * BN : 0;
    constexpr int TABF = GN ? 2 * BM : 0;
    static_assert(SMEM % 4 == 0 && SSLF % 4 == 0, "16-byte aligned LDS regions");
    __shared__ __attribute__((aligned(16))) float smem[SMEM + SSLF + TABF];
    // (scale, shift) rows: written once at kernel start, read by the store loop; per-row scene slot / scale-shift row
    float* const ssl = smem + SMEM;
    int* const rowsc = reinterpret_cast<int*>(smem + SMEM + SSLF);
    int* const rowss = rowsc + BM;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;

    const int z = blockIdx.y;
    int bid = blockIdx.x;
    {
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
    // IL == 2: operand tiles go from global memory straight into the swizzled LDS stage (buffer_load_dwordx4 ... lds, 1 KiB = 8
    // rows x 128 B per wave instruction): no staging registers, no ds_write pass, no per-tile address arithmetic -- a wave's
    // chunks have fixed per-lane byte offsets (voffset) and the K position moves in the scalar offset.  The LDS image is
    // lane-linear, so the XOR swizzle is applied on the global side: lane (row r, slot q') fetches k-quad q' ^ ((r >> 1) & 7).
    constexpr int CH = (BM + BN) / 8;             // 1-KiB chunks per tile
    constexpr int NI = (CH + NW - 1) / NW;        // chunks per wave (a ragged last round re-fetches chunks 0.. : same bytes)
    int dvoff[IL == 2 ? NI : 1];
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const float* const xb1 = a1 + row0 * p.lda1;
    const float* const xb2 = a2 ? a2 + row0 * p.lda2 : xb1;
    const float* const wb = w + (int64_t)col0 * p.ldw;
    if constexpr (IL == 2) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int c = wave_u + NW * i;
            if (c >= CH) c -= CH;
            const int q = (lane & 7) ^ ((((c & 1) << 2) + (lane >> 4)) & 7);
            const bool isx = c < BM / 8;
            const int r = (isx ? c : c - BM / 8) * 8 + (lane >> 3);
            const bool ok = isx ? r < rows_here : col0 + r < p.n;
            dvoff[i] = (ok ? r : 0) * (int)(isx ? p.lda1 : p.ldw) * 4 + q * 16;          // host guarantees lda1 == lda2
        }
    }
    auto dma_tile = [&](int kt, float* stage) {
        if constexpr (IL == 2) {
            const int k0 = kt * BK;
            const bool seg1 = k0 < p.k1;
            const float* const xb = seg1 ? xb1 : xb2;
            const int sx = (seg1 ? k0 : k0 - p.k1) * 4, sw = k0 * 4;
            __attribute__((address_space(3))) char* lbase = (__attribute__((address_space(3))) char*)stage;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                int c = wave_u + NW * i;
                if (c >= CH) c -= CH;
                const bool isx = c < BM / 8;               // wave-uniform: scalar selects, no branch
#if defined(__HIP_DEVICE_COMPILE__)                        // (the host pass of hipcc has no buffer-resource builtins)
                const __amdgpu_buffer_rsrc_t rs =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(isx ? xb : wb), 0, 0x7fffffff, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lbase + c * 1024, 16, dvoff[i], isx ? sx : sw, 0, 0);
#else
                (void)isx; (void)sx; (void)sw; (void)xb; (void)lbase;
#endif
            }
        }
    };
    if constexpr (IL == 2) dma_tile(0, smem);
    else load_tile(0);
    // per-row tables of the GroupNorm epilogue (scene slot, scale/shift row): built here, under the first operand tile's
    // latency, so that the epilogue has no integer division, no conditioning-mode branches and one barrier fewer
    if constexpr (GN) {
        for (int t = tid; t < BM; t += T) {
            const int sc = t / N;
            const bool ok = t < rows_here;
            rowsc[t] = ok ? sc : 0;
            // PER_SCENE / BY_INDEX rows are staged in ssl (the host only picks tiles with <= SSL_MAX scenes for those modes)
            rowss[t] = !ok ? 0 : (p.ss_mode == DSC_SS_PER_SLOT ? t - sc * N : (int)(row0 + t));
        }
    }
    if constexpr (GN) {
        if (ss_lds) {
            // all index loads, then all value loads, then the LDS writes: three unrolled passes so the round trips overlap
            int64_t srow[SSV];
            int soff[SSV];
#pragma unroll
            for (int j = 0; j < SSV; ++j) {
                const int f = tid + T * j;                     // (scene, half, column) flattened
                const int sc = f / (2 * BN), hc = f % (2 * BN);
                const int scc = sc < scenes_here ? sc : 0;     // slots past the block's scenes re-stage scene 0 (never read)
                srow[j] = (p.ss_mode == DSC_SS_PER_SCENE) ? scene0 + scc : dsc_clamp_index(p.ss_index[scene0 + scc], p.ss_rows);
                soff[j] = (hc >= BN ? p.n : 0) + col0 + (hc % BN);
            }
            float sval[SSV];
#pragma unroll
            for (int j = 0; j < SSV; ++j) sval[j] = p.scale_shift[srow[j] * p.ld_ss + soff[j]];
#pragma unroll
            for (int j = 0; j < SSV; ++j) ssl[tid + T * j] = sval[j];
        }
    }
    // IL kernels start the accumulators at the bias (the loads land under the first operand tile) instead of adding it after
    // the main loop, where 16 dependent L2 round trips sat exposed at the top of the epilogue
    {
        const bool bvec = IL && bias && (col0 + BN <= p.n) && ((reinterpret_cast<uintptr_t>(bias) & 15) == 0);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = col0 + (wn * TN + tn) * 32 + 8 * q + 4 * half;
                f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                if (bvec) b4 = *reinterpret_cast<const f32x4*>(bias + c);
                else if (IL && bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float bv = bias[c + e < p.n ? c + e : 0];        // clamped address: no branch per element
                        b4[e] = c + e < p.n ? bv : 0.0f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) acc[tm][tn][4 * q + e] = b4[e];
            }
    }
    // GroupNorm affine of this lane's 4 columns per column tile (epilogue layout): 8 registers per tile, loaded once here
    f32x4 gab[GN ? 2 * TN : 1];
    if constexpr (GN) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int c = col0 + (wn * TN + tn) * 32 + (lane & 7) * 4;
            gab[2 * tn] = *reinterpret_cast<const f32x4*>(p.gamma + c);
            gab[2 * tn + 1] = *reinterpret_cast<const f32x4*>(p.beta + c);
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
    if constexpr (IL == 2) {
        static_assert(BK == 32, "IL: BK = 32 (8 quads per row)");
        constexpr int S = BK / 8;
        constexpr int MM = TM * TN * 4;           // MFMAs per group
        static_assert(S == 4 && MM >= TM + TN + NI, "IL: the MFMA groups are too short to carry the staging instructions");
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
        auto mma = [&](const f32x4 (&xf)[TM], const f32x4 (&wf)[TN], int j0, int j1) {
#pragma unroll
            for (int j = j0; j < j1; ++j)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[tn][j], xf[tm][j], acc[tm][tn], 0, 0, 0);
        };
        __builtin_amdgcn_s_waitcnt(0x0f70);       // vmcnt(0): tile 0 is in LDS stage 0 (this wave's chunks)
        __syncthreads();
        DSC_STAMP(1);
        frags(smem, 0, xfA, wfA);
        for (int kt = 0; kt < nk; ++kt) {
            const float* cur = smem + (kt & 1) * IL_STAGE;
            float* nxt = smem + ((kt + 1) & 1) * IL_STAGE;
            __builtin_amdgcn_sched_barrier(0);
            // group 0: MFMAs(A), the fragment reads of group 1, the DMA of tile kt+1 into the other stage (its last reader
            // passed the barrier of tile kt-1)
            frags(cur, 1, xfB, wfB);
            if constexpr (PROBE != 1) dma_tile(kt + 1 < nk ? kt + 1 : nk - 1, nxt);
            mma(xfA, wfA, 0, 4);
#pragma unroll
            for (int i = 0; i < TM + TN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, MM - (TM + TN) - NI, 0);
            __builtin_amdgcn_sched_barrier(0);
            // groups 1, 2: MFMAs with the fragment reads of the following group
            frags(cur, 2, xfA, wfA);
            mma(xfB, wfB, 0, 4);
#pragma unroll
            for (int i = 0; i < TM + TN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, MM - 2 * (TM + TN), 0);
            __builtin_amdgcn_sched_barrier(0);
            frags(cur, 3, xfB, wfB);
            mma(xfA, wfA, 0, 4);
#pragma unroll
            for (int i = 0; i < TM + TN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, MM - 2 * (TM + TN), 0);
            __builtin_amdgcn_sched_barrier(0);
            // group 3: MFMAs(B); in its middle: this wave's DMA retired (vmcnt), the block barrier (every wave's chunks of
            // tile kt+1 are in LDS, every read of tile kt has returned), then the first fragment reads of tile kt+1
            mma(xfB, wfB, 0, 2);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0), lgkmcnt / expcnt untouched
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            frags(nxt, 0, xfA, wfA);
            mma(xfB, wfB, 2, 4);
#pragma unroll
            for (int i = 0; i < TM + TN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, MM / 2 - (TM + TN), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    } else if constexpr (IL == 1) {
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
            if constexpr (PROBE != 1 && PROBE != 2) stage_tile(nxt);
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
            if constexpr (PROBE != 1 && PROBE != 2) load_tile(kt + 2 < nk ? kt + 2 : nk - 1);
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
            if constexpr (PROBE != 2) __syncthreads();
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
    // bias (classic loops; the IL kernels started from it)
    if (!IL && bias) {
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
        f32x2* stat = reinterpret_cast<f32x2*>(smem + 2 * CT * BM);   // [spt*G] (mean, rstd); spt * G <= 80 for N >= 4
        const int nstat = scenes_here * G;
        const float inv_cnt = 1.0f / (64.0f * (float)N);
        // One pass over the accumulators (Chan's pairwise update): every lane reduces its 16 channels to (sum, centred sum
        // of squares), the two lane halves are merged, then one wave per (scene, group) merges the 2N token entries about
        // their common mean.  No E[x^2] - E[x]^2 cancellation anywhere: same quality as mean-then-variance, half the passes.
        // Two-wide vector types: v_pk_add_f32 / v_pk_fma_f32 halve the VALU instruction count of this block.
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int tl = (wm * TM + tm) * 32 + l31;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                f32x2 s2 = {0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 16; r += 2) s2 += f32x2{acc[tm][tn][r], acc[tm][tn][r + 1]};
                const float s16 = s2[0] + s2[1];
                const float m16 = s16 * (1.0f / 16.0f);
                const f32x2 m2 = {m16, m16};
                f32x2 q2 = {0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 d = f32x2{acc[tm][tn][r], acc[tm][tn][r + 1]} - m2;
                    q2 += d * d;
                }
                const float q16 = q2[0] + q2[1];
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
            const float mu = wave_sum_dpp(s) * inv_cnt;
            float q = 0.f;
            for (int j = lane; j < 2 * N; j += 64) {
                const int ct = 2 * g + (j >= N ? 1 : 0);
                const int tk = (j >= N ? j - N : j);
                const float d = P[ct * BM + sc * N + tk] * (1.0f / 32.0f) - mu;
                q += Q[ct * BM + sc * N + tk] + 32.0f * d * d;
            }
            q = wave_sum_dpp(q);
            if (lane == 0) stat[st] = f32x2{mu, 1.0f / sqrtf(q * inv_cnt + p.eps)};
        }
        __syncthreads();
        DSC_STAMP(7);
        DSC_STAMP(3);
        float* zp = p.preact ? p.preact + (int64_t)z * p.sy : nullptr;
        const bool yfast = (p.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
        // normalise, affine, scale/shift, SiLU, residual, store -- in the transposed (row-major) patch layout
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int g = (wn * TN + tn) >> 1;
            const int c = col0 + (wn * TN + tn) * 32 + cq * 4;
            const f32x4 ga = gab[2 * tn], be = gab[2 * tn + 1];
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
                        const int sc = rowsc[tl];
                        const f32x2 mr = stat[sc * G + g];
                        f32x4 v = *reinterpret_cast<const f32x4*>(patch + (tr + 8 * i) * TLD + cq * 4);
                        if (zq) *reinterpret_cast<f32x4*>(zq) = v;
                        // (v - mu) * rstd * gamma + beta  =  v * A + B,  A = rstd * gamma,  B = beta - mu * A  (packed pairs)
                        const f32x2 rs2 = {mr[1], mr[1]}, mu2 = {mr[0], mr[0]};
                        f32x2 v01 = {v[0], v[1]}, v23 = {v[2], v[3]};
                        const f32x2 a01 = rs2 * f32x2{ga[0], ga[1]}, a23 = rs2 * f32x2{ga[2], ga[3]};
                        v01 = v01 * a01 + (f32x2{be[0], be[1]} - mu2 * a01);
                        v23 = v23 * a23 + (f32x2{be[2], be[3]} - mu2 * a23);
                        if (has_ss) {
                            // rows not staged in LDS: per-slot / per-token tables, or (tiles with more scenes than ssl has
                            // slots -- the host avoids them) the scene's row straight from the global table
                            const float* ss = ss_lds ? ssl + sc * (2 * BN) + (wn * TN + tn) * 32 + cq * 4
                                : ssb + (p.ss_mode == DSC_SS_PER_SCENE ? scene0 + sc
                                         : p.ss_mode == DSC_SS_BY_INDEX ? dsc_clamp_index(p.ss_index[scene0 + sc], p.ss_rows) : (int64_t)rowss[tl]) * p.ld_ss;
                            const f32x4 sc4 = *reinterpret_cast<const f32x4*>(ss);
                            const f32x4 sh4 = *reinterpret_cast<const f32x4*>(ss + (ss_lds ? BN : p.n));
                            v01 = v01 * (f32x2{sc4[0], sc4[1]} + 1.0f) + f32x2{sh4[0], sh4[1]};
                            v23 = v23 * (f32x2{sc4[2], sc4[3]} + 1.0f) + f32x2{sh4[2], sh4[3]};
                        }
                        v = f32x4{v01[0], v01[1], v23[0], v23[1]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = dsc_silu_fast(v[e]);
                        if (res) {
                            if (EPF && use_pre) {
                                v += rpre[(tn * TM + tm) * 4 + i];
                            } else if (rfast) {
                                v += *reinterpret_cast<const f32x4*>(res + (row0 + tl) * p.ldr + c);
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
        // LNF: per-token mean / variance over the n channels = over this lane's TN * 16 registers, its partner half (lane ^ 32)
        // and the WN waves of the block (two passes: mean, then the centred sum of squares -- as the reference / ATen do)
        float ln_mu[LNF ? TM : 1], ln_rs[LNF ? TM : 1];
        f32x4 ln_g[LNF ? TN : 1];
        if constexpr (LNF) {
            static_assert(!GN && WM == 1, "LNF: one row of waves spans all output channels");
            float* Pl = smem;                       // [WN][BM]
            const float inv_n = 1.0f / (float)p.n;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    float s_ = 0.f;
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float d = pass ? acc[tm][tn][r] - ln_mu[tm] : acc[tm][tn][r];
                            s_ += pass ? d * d : d;
                        }
                    s_ += __shfl_xor(s_, 32, 64);
                    if (half == 0) Pl[wn * BM + tm * 32 + l31] = s_;
                }
                __syncthreads();
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    float t_ = 0.f;
#pragma unroll
                    for (int w_ = 0; w_ < WN; ++w_) t_ += Pl[w_ * BM + tm * 32 + l31];
                    if (pass == 0) ln_mu[tm] = t_ * inv_n;
                    else ln_rs[tm] = 1.0f / sqrtf(t_ * inv_n + p.eps);
                }
                __syncthreads();
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                ln_g[tn] = *reinterpret_cast<const f32x4*>(p.gamma + col0 + (wn * TN + tn) * 32 + cq * 4);
        }
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
                    for (int e = 0; e < 4; ++e)
                        v[e] = LNF ? (acc[tm][tn][4 * q + e] - ln_mu[tm]) * ln_rs[tm] : dsc_act(acc[tm][tn][4 * q + e], ACT);
                    *reinterpret_cast<f32x4*>(patch + l31 * TLD + 8 * q + 4 * half) = v;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int tl = tl0 + tm * 32 + 8 * i;
                    if (tl < rows_here) {
                        f32x4 v = *reinterpret_cast<const f32x4*>(patch + (tr + 8 * i) * TLD + cq * 4);
                        if constexpr (LNF) v *= ln_g[tn];
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

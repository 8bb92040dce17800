// K-parallel fp32 MFMA GEMM for launches too small to fill the chip (round 6).
//
// The reference's own generation call shape is ONE scene per call (scripts/generate_diffusion.py:314-323: generate_layout(batch_size=1),
// 12 or 21 objects), and three of the five BASELINE configurations carry 128 scenes of 12 objects.  There the tile kernels of
// gemm_core.h make 8..100 blocks that each walk K serially: one global -> LDS -> register round trip per 32-wide K tile, ~1 us each,
// 18 us for a K = 512 launch whatever the (tiny) amount of arithmetic -- 56 GroupNorm launches of 18 us are 58 % of a B = 1 reverse step.
//
// This kernel turns the K walk sideways.  A block owns <= 32 (or <= 16: R16 below) token rows (whole scenes) x 64 output channels (exactly one GroupNorm
// group) and splits K over its EIGHT waves: wave w owns K slice [w K/8, (w+1) K/8) and brings it to the matrix cores on its own -- no
// block barrier in the product, the loads of a whole K <= 512 slice set in flight at once (two forms of the K section, below: through a
// wave-private LDS image, or fragments straight from global memory).  The eight partial 32 x 32 tiles meet in LDS, are summed in a fixed order
// (wave 0..7) by the whole block in ROW layout (thread = token row x 16-byte channel quad), and the epilogue -- bias, activation /
// GroupNorm + (scale, shift) + SiLU, residual, saved pre-activation -- runs in that layout with fully coalesced global accesses.
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains, as gemm_core.h); the K sum is associated as 8 slice sums instead of one
// chain (error <= the single chain's).  Same C ABI, same argument struct, same epilogue semantics as dsc_gemm::gemm_kernel; the
// dispatch (gemm_mfma.hip: skinny_plan) takes a launch only when ALL its blocks fit one round of the 256 CUs.
//
// Operand fragments without LDS (forms 1, 2, 4, 0): a lane (l & 31 = row of the 32-row tile, l >> 5 = half) reads 16 bytes = 4 consecutive k at
// k = base + 8 u + 4 half; MFMA step e of unit u then multiplies element e of both operands' quads -- the two halves supply k = base +
// 8 u + e and base + 8 u + 4 + e.  A and W use the same rule, so every k of the slice meets its partner exactly once (the ORDER of the k
// inside a unit differs from the tile kernels'; the set does not).
#pragma once
#include "dsc_common.h"

// tools/skinny_probe.hip defines these to take s_memtime stamps of one wave; the product build compiles them away
#ifndef DSC_SKINNY_STAMP
#define DSC_SKINNY_STAMP(id)
#define DSC_SKINNY_STAMP_LOADED
#define DSC_SKINNY_STAMP_END
#define DSC_SKINNY_STAMP_DECL
#endif

namespace dsc_skinny {

constexpr int NW = 8;          // waves per block = K slices
constexpr int BN = 64;         // output channels per block = one GroupNorm group
constexpr int TLD = 36;        // padded row (floats) of a 32 x 32 partial tile in LDS
constexpr int UG = 4;          // units (of 8 k) per register buffer: 32 k
constexpr int SCRATCH = 256;   // floats behind the partial tiles: row sums / scene statistics

constexpr int STAGED = 8;      // value of the NG template parameter that selects the LDS-staged K section
constexpr int SLD = 68;        // padded row (floats) of a staged 32-row x 64-k operand image: 256 + 16 bytes
constexpr int WAVE_LDS_STAGED = 2 * 32 * SLD;          // per wave: one weight image + the token image (floats)
constexpr int WAVE_LDS_DIRECT = 2 * 32 * TLD;          // per wave: its two partial tiles
template <int NG>
constexpr int wave_lds() { return NG == STAGED ? WAVE_LDS_STAGED : WAVE_LDS_DIRECT; }
template <int NG>
constexpr size_t lds_bytes() { return sizeof(float) * (size_t)(NW * wave_lds<NG>() + SCRATCH); }

// How a wave brings its K slice to the matrix cores (template parameter NG):
//   STAGED (K % 512 == 0: slices of whole 64-k chunks) -- the measured form for the launches that matter.  Reading fragments straight
//     from global memory costs 32 cache lines of 2 KB stride per instruction, 32 bytes used of each and every line touched by four
//     instructions: tools/skinny_probe shows the operands of a K = 512 launch arriving 4.3 us (first four waves) and 7 us (last four)
//     after the request -- 25..35 GB/s per CU -- with the 64 MFMAs of a wave taking 1.8 us.  Here the lanes read ALONG k instead (16
//     lanes x 16 bytes = the 256-byte chunk of a row, four rows per instruction: whole lines, each touched once), drop the quads into
//     a wave-private LDS image (padded rows) and pick their MFMA fragments out of it -- wave-level ordering only, no block barrier.
//     One weight image (32 channels x 64 k) + the token image per wave: the second channel tile overwrites the first once its
//     fragments are in registers; the partial tiles later reuse the same bytes.
//   1, 2, 4 -- fragments straight from global memory, the whole slice in NG register buffers of 32 k (K <= 256 NG), every load issued
//     before the first MFMA; 0 -- the same with two buffers in flight, any K.  (What is left for them: K = 128, 256, 384 ...)
//   R16 (staged form only): the block holds <= 16 token rows -- the 12-object bedroom scene of the reference's generation script -- and multiplies
//     with v_mfma_f32_16x16x4_f32: four 16-channel tiles x ONE 16-token tile per wave, half the MFMA time of the padded 32-row tile.  Lane
//     (l & 15 = row of the 16-row tile, l >> 4 = k quarter) picks 16 bytes at k = 16 u + 4 (l >> 4) out of the same LDS images; MFMA step e of
//     unit u multiplies element e of both quads (k = 16 u + 4 q + e for the four quarters q): same rule for both operands.
template <bool GN, int NG, bool R16 = false>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_kernel(const dsc_gemm_args p, int ncolblk, int rows_per_blk) {
    static_assert(!R16 || NG == STAGED, "16-row blocks: the LDS-staged form only");
    extern __shared__ float smem[];
    DSC_SKINNY_STAMP_DECL
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int z = blockIdx.y;
    const int rb = blockIdx.x / ncolblk, cb = blockIdx.x % ncolblk;
    const int64_t row0 = (int64_t)rb * rows_per_blk;
    const int rows_here = (int)((p.m - row0) < (int64_t)rows_per_blk ? (p.m - row0) : (int64_t)rows_per_blk);
    const int col0 = cb * BN;

    const float* a1 = p.a1 + (int64_t)z * p.sa1;
    const float* a2 = p.a2 ? p.a2 + (int64_t)z * p.sa2 : nullptr;
    const float* w = p.w + (int64_t)z * p.sw;
    const float* bias = p.bias ? p.bias + (int64_t)z * p.sbias : nullptr;
    const float* res = p.residual ? p.residual + (int64_t)z * p.sres : nullptr;
    float* y = p.y + (int64_t)z * p.sy;

    // ---- this wave's K slice (inside ONE operand segment: the host checks k1 % (K / 8) == 0 when there are two)
    const int K = p.k1 + p.k2;
    const int KS = K / NW;
    const int U = KS >> 3;                       // units of 8 k
    const int kbeg = wave * KS;
    const float* ab;
    int64_t lda;
    int ka;
    if (kbeg < p.k1) { ab = a1; lda = p.lda1; ka = kbeg; }
    else             { ab = a2; lda = p.lda2; ka = kbeg - p.k1; }
    // rows past the end read row 0 of the block: their accumulators are never stored
    const float* ap = ab + (row0 + (l31 < rows_here ? l31 : 0)) * lda + ka + 4 * half;
    const float* wp[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) wp[ct] = w + (int64_t)(col0 + ct * 32 + l31) * p.ldw + kbeg + 4 * half;

    f32x16 acc[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
    f32x4 acc16[4];                              // R16: [16-channel tile]: lane = token l & 15, element r = channel 4 (l >> 4) + r
#pragma unroll
    for (int t = 0; t < 4; ++t) acc16[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto load = [&](f32x4 (&xa)[UG], f32x4 (&xw)[2][UG], int g) {
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const int uu = g * UG + u;
            const int off = 8 * (uu < U ? uu : 0);         // a unit past the slice re-reads unit 0 (valid memory); it is not multiplied
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) xw[ct][u] = *reinterpret_cast<const f32x4*>(wp[ct] + off);
            xa[u] = *reinterpret_cast<const f32x4*>(ap + off);
        }
    };
    auto mma = [&](const f32x4 (&xa)[UG], const f32x4 (&xw)[2][UG], int g) {
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            if (g * UG + u < U) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(xw[ct][u][e], xa[u][e], acc[ct], 0, 0, 0);
            }
        }
    };

    // ---- row layout of the epilogue: thread = (token row r, channel quad cq).  Everything the epilogue reads that does not depend on the
    // product -- bias, GroupNorm affine, the (scale, shift) row, the residual quad -- is requested right behind the operand loads and lands
    // under the MFMAs: at this size a launch is a chain of memory latencies, and each one taken off the chain is ~1 us of a ~10 us launch
    const int cq = tid & 15, r = tid >> 4;
    const int c = col0 + cq * 4;
    const bool ok = r < rows_here;
    const int64_t tok = row0 + r;
    const int N = GN ? p.tokens_per_scene : 1;
    const int sc = (GN && ok) ? r / N : 0;
    const int64_t scene0 = GN ? row0 / N : 0;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f}, r4 = b4, ga = b4, be = b4, sc4 = b4, sh4 = b4;
    auto prefetch_epilogue = [&]() {
        if (bias) b4 = *reinterpret_cast<const f32x4*>(bias + c);
        if (res && ok) r4 = *reinterpret_cast<const f32x4*>(res + tok * p.ldr + c);
        if constexpr (GN) {
            ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
            be = *reinterpret_cast<const f32x4*>(p.beta + c);
            if (p.scale_shift && ok) {
                const int64_t srow = p.ss_mode == DSC_SS_PER_TOKEN ? tok
                                   : p.ss_mode == DSC_SS_PER_SCENE ? scene0 + sc
                                   : p.ss_mode == DSC_SS_PER_SLOT  ? (int64_t)(r % N)
                                   : dsc_clamp_index(p.ss_index[scene0 + sc], p.ss_rows);
                const float* ss = p.scale_shift + srow * p.ld_ss + c;
                sc4 = *reinterpret_cast<const f32x4*>(ss);
                sh4 = *reinterpret_cast<const f32x4*>(ss + p.n);
            }
        }
    };

    DSC_SKINNY_STAMP(0)
    float* wbase = smem + wave * wave_lds<NG>();           // this wave's bytes: staging images, then its partial tiles
    if constexpr (NG == STAGED) {
        const int rsub = lane >> 4, seg = lane & 15;        // coalesced: 16 lanes x 16 bytes = one row's 64-k chunk, 4 rows per instruction
        const float* gwp[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) gwp[ct] = w + (int64_t)(col0 + ct * 32 + rsub) * p.ldw + kbeg + seg * 4;
        const int64_t wstep = 4 * p.ldw;
        int64_t aoff[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rr = 4 * i + rsub;                    // rows past the end read row 0 of the block: their accumulators are never stored
            aoff[i] = (row0 + (rr < rows_here ? rr : 0)) * lda + ka + seg * 4;
        }
        f32x4 qa[8], qw[2][8];
        auto gload_a = [&](int ch) {
#pragma unroll
            for (int i = 0; i < (R16 ? 4 : 8); ++i) qa[i] = *reinterpret_cast<const f32x4*>(ab + aoff[i] + 64 * ch);
        };
        auto gload_w = [&](int ct, int ch) {
#pragma unroll
            for (int i = 0; i < 8; ++i) qw[ct][i] = *reinterpret_cast<const f32x4*>(gwp[ct] + i * wstep + 64 * ch);
        };
        float* wimg = wbase;                                // [32 channels][SLD]
        float* aimg = wbase + 32 * SLD;                     // [32 tokens][SLD]
        const int st_off = rsub * SLD + seg * 4;            // staging write: row 4 i + rsub, 16-byte piece seg
        const int fr_off = l31 * SLD + 4 * half;            // fragment read: row l31, k = 8 u + 4 half
        const int nch = KS >> 6;
        gload_w(0, 0);
        gload_a(0);
        gload_w(1, 0);
        prefetch_epilogue();
        DSC_SKINNY_STAMP_LOADED
        if constexpr (R16) {
            const int l15 = lane & 15, kq = lane >> 4;
            const int f16 = l15 * SLD + 4 * kq;             // fragment read: row l15 of a 16-row tile, k = 16 u + 4 kq
            auto mma16 = [&](int t0, const f32x4 (&fw)[2][4], const f32x4 (&fa)[4]) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int t2 = 0; t2 < 2; ++t2)
                            acc16[t0 + t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[t2][u][e], fa[u][e], acc16[t0 + t2], 0, 0, 0);
            };
            for (int ch = 0; ch < nch; ++ch) {
                f32x4 fa[4], fw[2][4];
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(wimg + 4 * i * SLD + st_off) = qw[0][i];
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(aimg + 4 * i * SLD + st_off) = qa[i];      // token rows 0..15 only
                if (ch + 1 < nch) { gload_w(0, ch + 1); gload_a(ch + 1); }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    fa[u] = *reinterpret_cast<const f32x4*>(aimg + f16 + 16 * u);
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) fw[t2][u] = *reinterpret_cast<const f32x4*>(wimg + t2 * 16 * SLD + f16 + 16 * u);
                }
                mma16(0, fw, fa);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(wimg + 4 * i * SLD + st_off) = qw[1][i];
                if (ch + 1 < nch) gload_w(1, ch + 1);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) fw[t2][u] = *reinterpret_cast<const f32x4*>(wimg + t2 * 16 * SLD + f16 + 16 * u);
                mma16(2, fw, fa);
                __builtin_amdgcn_wave_barrier();
            }
        } else
        for (int ch = 0; ch < nch; ++ch) {
            f32x4 fa[8], fw[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(wimg + 4 * i * SLD + st_off) = qw[0][i];
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(aimg + 4 * i * SLD + st_off) = qa[i];
            if (ch + 1 < nch) { gload_w(0, ch + 1); gload_a(ch + 1); }       // the registers just staged take the next chunk
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                fw[u] = *reinterpret_cast<const f32x4*>(wimg + fr_off + 8 * u);
                fa[u] = *reinterpret_cast<const f32x4*>(aimg + fr_off + 8 * u);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[u][e], fa[u][e], acc[0], 0, 0, 0);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(wimg + 4 * i * SLD + st_off) = qw[1][i];
            if (ch + 1 < nch) gload_w(1, ch + 1);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 8; ++u) fw[u] = *reinterpret_cast<const f32x4*>(wimg + fr_off + 8 * u);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[u][e], fa[u][e], acc[1], 0, 0, 0);
            __builtin_amdgcn_wave_barrier();
        }
    } else if constexpr (NG > 0) {
        f32x4 xa[NG][UG], xw[NG][2][UG];
#pragma unroll
        for (int g = 0; g < NG; ++g) load(xa[g], xw[g], g);
        if constexpr (NG < 4) prefetch_epilogue();
        DSC_SKINNY_STAMP_LOADED
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            mma(xa[g], xw[g], g);
            if constexpr (NG >= 4) {
                if (g == 1) prefetch_epilogue();        // four buffers + the epilogue operands do not fit 256 registers: request them once two buffers are free
            }
        }
    } else {
        const int ng = (U + UG - 1) / UG;
        f32x4 xa0[UG], xw0[2][UG], xa1[UG], xw1[2][UG];
        load(xa0, xw0, 0);
        prefetch_epilogue();
        for (int g = 0; g < ng; g += 2) {
            if (g + 1 < ng) load(xa1, xw1, g + 1);
            mma(xa0, xw0, g);
            if (g + 2 < ng) load(xa0, xw0, g + 2);
            if (g + 1 < ng) mma(xa1, xw1, g + 1);
        }
    }

    DSC_SKINNY_STAMP(1)
    // ---- the eight partial tiles meet in LDS: wave w's bytes, [ct][token][channel]; MFMA layout: lane = token l31, acc[4 q + e] = channel
    // 8 q + 4 half + e  (R16: [token 16][64 channels], padded rows of SLD floats)
    if constexpr (R16) {
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(wbase + (lane & 15) * SLD + t * 16 + 4 * (lane >> 4)) = acc16[t];
    } else {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            float* t = wbase + ct * (32 * TLD) + l31 * TLD + 4 * half;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4*>(t + 8 * q) = f32x4{acc[ct][4 * q], acc[ct][4 * q + 1], acc[ct][4 * q + 2], acc[ct][4 * q + 3]};
        }
    }
    __syncthreads();
    DSC_SKINNY_STAMP(2)

    // ---- fixed summation order over the waves
    f32x4 v;
    {
        const float* t = R16 ? smem + (r & 15) * SLD + cq * 4 : smem + (cq >> 3) * (32 * TLD) + r * TLD + (cq & 7) * 4;
        v = *reinterpret_cast<const f32x4*>(t);
#pragma unroll
        for (int wv = 1; wv < NW; ++wv) v += *reinterpret_cast<const f32x4*>(t + wv * wave_lds<NG>());
    }
    v += b4;
    DSC_SKINNY_STAMP(3)

    if constexpr (!GN) {
        if (ok) {
            if (p.act_out != DSC_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = dsc_act(v[e], p.act_out);
            }
            if (res) v += r4;
            *reinterpret_cast<f32x4*>(y + tok * p.ldy + c) = v;
        }
    } else {
        // GroupNorm over (tokens of a scene) x (the 64 channels of this block), two passes (mean, then the centred sum of squares)
        float* S = smem + NW * wave_lds<NG>();         // [32] per-row sums (pass 1) / centred sums of squares (pass 2)
        float* MU = S + 64;                            // [8] per-scene mean
        float* RS = MU + 16;                           // [8] per-scene 1 / sqrt(var + eps)
        const int spt = rows_here / N;
        const float inv_cnt = 1.0f / (64.0f * (float)N);
        auto row16 = [](float x) {                     // sum over the 16 lanes (channel quads) of a token row
            x += __shfl_xor(x, 1, 64);
            x += __shfl_xor(x, 2, 64);
            x += __shfl_xor(x, 4, 64);
            x += __shfl_xor(x, 8, 64);
            return x;
        };
        if (p.preact && ok) *reinterpret_cast<f32x4*>(p.preact + (int64_t)z * p.sy + tok * p.ld_preact + c) = v;
        {
            const float s = row16(ok ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f);
            if (cq == 0) S[r] = s;
        }
        __syncthreads();
        for (int st = wave; st < spt; st += NW) {
            float s = lane < N ? S[st * N + lane] : 0.f;          // N <= 32 rows per scene
            s = wave_sum_dpp(s);
            if (lane == 0) MU[st] = s * inv_cnt;
        }
        __syncthreads();
        const float mu = MU[sc];
        {
            float q = 0.f;
            if (ok) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[e] - mu; q += d * d; }
            }
            q = row16(q);
            if (cq == 0) S[32 + r] = q;
        }
        __syncthreads();
        for (int st = wave; st < spt; st += NW) {
            float q = lane < N ? S[32 + st * N + lane] : 0.f;
            q = wave_sum_dpp(q);
            if (lane == 0) RS[st] = 1.0f / sqrtf(q * inv_cnt + p.eps);
        }
        __syncthreads();
        if (ok) {
            const float rs = RS[sc];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = rs * ga[e];
                o[e] = v[e] * a + (be[e] - mu * a);                   // (z - mu) rstd gamma + beta, as gemm_core.h
            }
            if (p.scale_shift) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = o[e] * (sc4[e] + 1.0f) + sh4[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = dsc_silu_fast(o[e]);
            if (res) o += r4;
            *reinterpret_cast<f32x4*>(y + tok * p.ldy + c) = o;
        }
    }
    DSC_SKINNY_STAMP_END
}

}  // namespace dsc_skinny

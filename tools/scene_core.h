// EXPERIMENT (not part of libdiffuscene_hip): scene-resident layer kernels measured with tools/scene_tune.py; results and
// conclusions in DESIGN.md section 7.  Kept as groundwork for the persistent per-scene kernel.
// Scene-resident building blocks (gfx950): ONE workgroup of 512 threads owns ONE scene (N <= 80 tokens) and computes a
// whole layer for it -- D[channel][token] = sum_k W[channel][k] X[token][k] over all n <= 512 output channels -- so that a
// persistent kernel can walk the layer list of the denoiser for its scene without kernel boundaries or inter-block
// synchronisation (scenes are independent: GroupNorm, attention and LayerNorm never cross a scene).
//
// Tiling: v_mfma_f32_16x16x4_f32 (80 tokens = 5 x 16, no padding); wave w owns channels [64w, 64w+64) = 4 channel tiles
// x 5 token tiles = 20 accumulator quads (80 VGPRs).  For GroupNorm(8 groups of 64 channels) a wave therefore holds one
// complete (scene, group): mean / variance are wave-local (no LDS, no barrier).
// K is staged 16 wide, double-buffered in LDS (rows padded to 24 floats: conflict-free ds_read_b128 for the 16x16x4
// fragment pattern -- lane (i = l%16, g = l/16) reads row i, floats [4g, 4g+4) = the k slots of 4 consecutive MFMAs; both
// operands use the same k permutation).  One barrier per K tile; the next tile's global loads are in flight during the
// MFMAs of the current one.
#pragma once
#include <type_traits>
#include "../diffuscene_amd/csrc/dsc_common.h"

namespace dsc_scene {

constexpr int SC_T = 512;              // threads per scene block
constexpr int SC_ROWS = 80;            // max tokens per scene handled by this core
constexpr int SC_TT = 5;               // token tiles of 16
constexpr int SC_BK = 16;
constexpr int SC_LD = 24;              // padded LDS row
constexpr int SC_NMAX = 512;           // output channels per pass
constexpr int SC_STAGE = (SC_NMAX + SC_ROWS) * SC_LD;       // floats per LDS stage
constexpr int SC_SMEM_FLOATS = 2 * SC_STAGE;                // 28416 floats = 113.7 KB

// One layer for one scene.  `p` is the batch-level descriptor of the product GEMM (include/diffuscene_hip.h); rows
// [scene*N, scene*N + N) of every matrix belong to this block.  GN selects the fused GroupNorm+(scale,shift)+SiLU epilogue
// (p.n must be 512).  All 512 threads must call this; ends with the block's stores issued (caller fences/syncs).
constexpr int SC_WP_WAVE_FLOATS = (64 + SC_ROWS) * SC_LD;    // wave-private LDS region: 64 weight rows + 80 token rows
constexpr int SC_WP_SMEM_FLOATS = 8 * SC_WP_WAVE_FLOATS;     // 27648 floats = 108 KB
constexpr int SC_G2_SMEM_FLOATS = 2 * 2 * (256 + SC_ROWS) * SC_LD + 64;   // two groups x two stages + barrier counters = 126 KB

// MODE 0: block-staged LDS tiles + one __syncthreads per K group; 1: fragments straight from global memory (uncoalesced
// in lane order, TA-bound -- kept for reference); 2: wave-private LDS staging, no block barrier in the main loop.
template <bool GN, int MODE = 2, int PROBE = 0>    // PROBE (tuning only): 1 = no loads in the loop, 2 = k never advances
__device__ __forceinline__ void scene_gemm(const dsc_gemm_args& p, const int scene, const int N, float* smem) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int64_t row0 = (int64_t)scene * N;
    const int n = p.n;
    const int nk = (p.k1 + p.k2) / SC_BK;

    f32x4 acc[SC_TT][4];
#pragma unroll
    for (int tt = 0; tt < SC_TT; ++tt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[tt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  if constexpr (MODE == 1) {
    // Barrier-free main loop: every wave fetches its own MFMA fragments straight from L2/L1 into registers -- lane
    // (li, lg) loads W[c0 + li][k0 + 4 lg .. +3] and X[t0 + li][k0 + 4 lg .. +3] (the 16x16x4 operand shape, the k slots
    // of 4 consecutive MFMAs) -- one 16-wide K group ahead of the MFMAs that consume it.  No LDS, no __syncthreads: the
    // 8 waves of the block drift freely, so one wave's fragment waits and epilogue overlap the other wave's MFMAs.
    (void)smem;
    const float* wp[4];
    const float* xp1[SC_TT];
    const float* xp2[SC_TT];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int c = wave * 64 + ct * 16 + li;
        wp[ct] = p.w + (int64_t)(c < n ? c : 0) * p.ldw + lg * 4;
    }
#pragma unroll
    for (int tt = 0; tt < SC_TT; ++tt) {
        const int t = tt * 16 + li;
        const int64_t r = row0 + (t < N ? t : 0);
        xp1[tt] = p.a1 + r * p.lda1 + lg * 4;
        xp2[tt] = p.a2 ? p.a2 + r * p.lda2 + lg * 4 : xp1[tt];
    }
    f32x4 wfA[4], xfA[SC_TT], wfB[4], xfB[SC_TT];
    auto fetch = [&](int kt, f32x4 (&wf)[4], f32x4 (&xf)[SC_TT]) {
        if (PROBE == 1 && kt > 1) return;
        const int k0 = (PROBE == 2) ? 0 : kt * SC_BK;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) wf[ct] = *reinterpret_cast<const f32x4*>(wp[ct] + k0);
        if (k0 < p.k1) {
#pragma unroll
            for (int tt = 0; tt < SC_TT; ++tt) xf[tt] = *reinterpret_cast<const f32x4*>(xp1[tt] + k0);
        } else {
#pragma unroll
            for (int tt = 0; tt < SC_TT; ++tt) xf[tt] = *reinterpret_cast<const f32x4*>(xp2[tt] + (k0 - p.k1));
        }
    };
    auto mma = [&](const f32x4 (&wf)[4], const f32x4 (&xf)[SC_TT]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int tt = 0; tt < SC_TT; ++tt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    acc[tt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ct][s], xf[tt][s], acc[tt][ct], 0, 0, 0);
    };
    // branch-free two-stage pipeline (nk is even: K is a multiple of 32); the last prefetch re-reads the final group
    fetch(0, wfA, xfA);
    for (int kt = 0; kt < nk; kt += 2) {
        fetch(kt + 1, wfB, xfB);
        __builtin_amdgcn_sched_barrier(0);
        mma(wfA, xfA);
        __builtin_amdgcn_sched_barrier(0);
        fetch(kt + 2 < nk ? kt + 2 : nk - 1, wfA, xfA);
        __builtin_amdgcn_sched_barrier(0);
        mma(wfB, xfB);
        __builtin_amdgcn_sched_barrier(0);
    }
    // rows >= N / channels >= n were fetched from row 0 / channel 0: their accumulators are never stored
  } else if constexpr (MODE == 3) {
    // Fragment-major operands: weights and activations are stored in the order the MFMA consumes them --
    //   Wf[channel tile c/16][K group k/16][lane = 16*((k%16)/4) + c%16][k%4],  Xf[scene][token tile][K group][lane][k%4]
    // so a fragment is ONE perfectly coalesced 1 KB load per wave (8 full lines), no LDS, no barrier, and the epilogue's
    // accumulator layout IS the next layer's operand layout (lane-contiguous 16-byte stores).
    (void)smem;
    const int KG = nk;                                                        // K groups of this layer (k1 + k2)
    const int KG1 = p.k1 / SC_BK;
    const float* wfp = p.w + ((int64_t)(wave * 4) * KG * 64 + lane) * 4;      // + (ct * KG + kg) * 256
    const float* xfp1 = p.a1 + ((int64_t)scene * SC_TT * KG1 * 64 + lane) * 4;  // + (tt * KG1 + kg) * 256
    const int KG2 = KG - KG1;
    const float* xfp2 = p.a2 ? p.a2 + ((int64_t)scene * SC_TT * KG2 * 64 + lane) * 4 : xfp1;
    f32x4 wfA[4], xfA[SC_TT], wfB[4], xfB[SC_TT];
    auto fetch = [&](int kg, f32x4 (&wf)[4], f32x4 (&xf)[SC_TT]) {
        if (PROBE == 1 && kg > 1) return;
        if (PROBE == 2) kg = 0;
        if (!(PROBE == 3 && kg > 1)) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) wf[ct] = *reinterpret_cast<const f32x4*>(wfp + ((int64_t)ct * KG + kg) * 256);
        }
        if (PROBE == 4 && kg > 1) return;
        if (kg < KG1) {
#pragma unroll
            for (int tt = 0; tt < SC_TT; ++tt) xf[tt] = *reinterpret_cast<const f32x4*>(xfp1 + ((int64_t)tt * KG1 + kg) * 256);
        } else {
#pragma unroll
            for (int tt = 0; tt < SC_TT; ++tt)
                xf[tt] = *reinterpret_cast<const f32x4*>(xfp2 + ((int64_t)tt * KG2 + (kg - KG1)) * 256);
        }
    };
    auto mma = [&](const f32x4 (&wf)[4], const f32x4 (&xf)[SC_TT]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int tt = 0; tt < SC_TT; ++tt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    acc[tt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ct][s], xf[tt][s], acc[tt][ct], 0, 0, 0);
    };
    // branch-free two-stage pipeline (KG is even: K is a multiple of 32); the last prefetch re-reads the final group
    fetch(0, wfA, xfA);
    for (int kg = 0; kg < KG; kg += 2) {
        fetch(kg + 1, wfB, xfB);
        __builtin_amdgcn_sched_barrier(0);
        mma(wfA, xfA);
        __builtin_amdgcn_sched_barrier(0);
        fetch(kg + 2 < KG ? kg + 2 : KG - 1, wfA, xfA);
        __builtin_amdgcn_sched_barrier(0);
        mma(wfB, xfB);
        __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (MODE == 4) {
    // Two independent 4-wave groups (one wave per SIMD each): group g owns channels [256g, 256g+256), stages ITS weight rows
    // and the scene's token rows into its own double-buffered LDS tiles (coalesced loads shared by the group's 4 waves, so
    // the bytes per MFMA match the product kernel) and synchronises with an LDS-counter barrier of its own.  The two groups
    // drift out of phase like two independent blocks on a CU: one group's staging / barrier / fragment-read gap is covered
    // by the other group's MFMAs.
    constexpr int GROWS = 256 + SC_ROWS;                      // staged rows per group and K group
    constexpr int GSTAGE = GROWS * SC_LD;                     // floats per stage
    const int grp = wave >> 2, gw = wave & 3, gt = tid & 255;
    float* gbase = smem + grp * (2 * GSTAGE);
    unsigned* bar = reinterpret_cast<unsigned*>(smem + 2 * 2 * GSTAGE) + grp * 32;   // one counter per group (own bank line)
    if (tid < 64) reinterpret_cast<unsigned*>(smem + 2 * 2 * GSTAGE)[tid] = 0u;
    __syncthreads();
    unsigned bar_target = 0;
    auto group_barrier = [&]() {
        bar_target += 4;
        if (lane == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < bar_target)
            __builtin_amdgcn_s_sleep(1);
    };
    // staging map of the group's 256 threads: 336 rows x 4 quads = 1344 quads -> 5.25 per thread
    constexpr int NQ = GROWS * 4;
    f32x4 st[6];
    auto gload = [&](int kt) {
        const int k0 = kt * SC_BK;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int f = gt + 256 * i;
            const int r = f >> 2, qd = f & 3;
            if (i < 5 || f < NQ) {
                if (r < 256) {
                    const int c = grp * 256 + r;
                    st[i] = *reinterpret_cast<const f32x4*>(p.w + (int64_t)(c < n ? c : 0) * p.ldw + k0 + qd * 4);
                } else {
                    const int t = r - 256;
                    const int64_t row = row0 + (t < N ? t : 0);
                    st[i] = (k0 < p.k1) ? *reinterpret_cast<const f32x4*>(p.a1 + row * p.lda1 + k0 + qd * 4)
                                        : *reinterpret_cast<const f32x4*>(p.a2 + row * p.lda2 + (k0 - p.k1) + qd * 4);
                }
            }
        }
    };
    auto lds_put = [&](float* stage) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int f = gt + 256 * i;
            const int r = f >> 2, qd = f & 3;
            if (i < 5 || f < NQ) *reinterpret_cast<f32x4*>(stage + r * SC_LD + qd * 4) = st[i];
        }
    };
    f32x4 wfA[4], xfA[SC_TT], wfB[4], xfB[SC_TT];
    auto lds_get = [&](const float* stage, f32x4 (&wf)[4], f32x4 (&xf)[SC_TT]) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
            wf[ct] = *reinterpret_cast<const f32x4*>(stage + (gw * 64 + ct * 16 + li) * SC_LD + lg * 4);
#pragma unroll
        for (int tt = 0; tt < SC_TT; ++tt)
            xf[tt] = *reinterpret_cast<const f32x4*>(stage + (256 + tt * 16 + li) * SC_LD + lg * 4);
    };
    auto mma = [&](const f32x4 (&wf)[4], const f32x4 (&xf)[SC_TT]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int tt = 0; tt < SC_TT; ++tt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    acc[tt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ct][s], xf[tt][s], acc[tt][ct], 0, 0, 0);
    };
    // the wave -> channel map of the epilogue below expects wave w to own channels [64w, 64w+64): grp*256 + gw*64 = 64*wave
    gload(0);
    lds_put(gbase);
    gload(1);
    group_barrier();
    lds_get(gbase, wfA, xfA);
    for (int kt = 0; kt < nk; kt += 2) {
        lds_put(gbase + GSTAGE);                              // group kt+1
        gload(kt + 2 < nk ? kt + 2 : nk - 1);
        group_barrier();
        lds_get(gbase + GSTAGE, wfB, xfB);
        __builtin_amdgcn_sched_barrier(0);
        mma(wfA, xfA);
        __builtin_amdgcn_sched_barrier(0);
        lds_put(gbase);                                       // group kt+2
        gload(kt + 3 < nk ? kt + 3 : nk - 1);
        group_barrier();
        lds_get(gbase, wfA, xfA);
        __builtin_amdgcn_sched_barrier(0);
        mma(wfB, xfB);
        __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (MODE == 2) {
    // Wave-private staging: each wave copies ITS 64 weight rows and the scene's token rows of the next K group into its
    // own LDS region with coalesced loads (4 lanes = 64 contiguous bytes of a row) and reads its MFMA fragments back with
    // conflict-free ds_read_b128.  Nothing is shared between waves, so the main loop has NO block barrier: LDS executes a
    // wave's instructions in order (write k+1 after the reads of k, read k+1 after its write), the global loads of group
    // k+2, the LDS write and the fragment reads of group k+1 are all issued ahead of the 80 MFMAs of group k.
    float* Wp = smem + wave * SC_WP_WAVE_FLOATS;
    float* Xp = Wp + 64 * SC_LD;
    const int sr = lane >> 2, sq = lane & 3;                  // staging: row sr + 16 i, quad sq
    const float* wg[4];
    const float* xg1[SC_TT];
    const float* xg2[SC_TT];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = wave * 64 + sr + 16 * i;
        wg[i] = p.w + (int64_t)(c < n ? c : 0) * p.ldw + sq * 4;
    }
#pragma unroll
    for (int i = 0; i < SC_TT; ++i) {
        const int t = sr + 16 * i;
        const int64_t r = row0 + (t < N ? t : 0);
        xg1[i] = p.a1 + r * p.lda1 + sq * 4;
        xg2[i] = p.a2 ? p.a2 + r * p.lda2 + sq * 4 : xg1[i];
    }
    f32x4 sw[4], sx[SC_TT];                                   // staging registers (one K group)
    f32x4 wfA[4], xfA[SC_TT], wfB[4], xfB[SC_TT];             // fragment registers (two K groups)
    auto gload = [&](int kt) {
        if (PROBE == 1 && kt > 1) return;
        const int k0 = (PROBE == 2) ? 0 : kt * SC_BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) sw[i] = *reinterpret_cast<const f32x4*>(wg[i] + k0);
        if (k0 < p.k1) {
#pragma unroll
            for (int i = 0; i < SC_TT; ++i) sx[i] = *reinterpret_cast<const f32x4*>(xg1[i] + k0);
        } else {
#pragma unroll
            for (int i = 0; i < SC_TT; ++i) sx[i] = *reinterpret_cast<const f32x4*>(xg2[i] + (k0 - p.k1));
        }
    };
    auto lds_put = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(Wp + (sr + 16 * i) * SC_LD + sq * 4) = sw[i];
#pragma unroll
        for (int i = 0; i < SC_TT; ++i) *reinterpret_cast<f32x4*>(Xp + (sr + 16 * i) * SC_LD + sq * 4) = sx[i];
        __builtin_amdgcn_wave_barrier();          // compiler ordering only; the LDS pipe keeps a wave's accesses in order
    };
    auto lds_get = [&](f32x4 (&wf)[4], f32x4 (&xf)[SC_TT]) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) wf[ct] = *reinterpret_cast<const f32x4*>(Wp + (ct * 16 + li) * SC_LD + lg * 4);
#pragma unroll
        for (int tt = 0; tt < SC_TT; ++tt) xf[tt] = *reinterpret_cast<const f32x4*>(Xp + (tt * 16 + li) * SC_LD + lg * 4);
        __builtin_amdgcn_wave_barrier();
    };
    auto mma = [&](const f32x4 (&wf)[4], const f32x4 (&xf)[SC_TT]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int tt = 0; tt < SC_TT; ++tt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    acc[tt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ct][s], xf[tt][s], acc[tt][ct], 0, 0, 0);
    };
    // branch-free pipeline (nk even): staging registers run two groups ahead, fragments one group ahead; the clamped
    // indices of the last iteration re-stage the final group (harmless)
    gload(0);
    lds_put();
    gload(1);
    lds_get(wfA, xfA);
    for (int kt = 0; kt < nk; kt += 2) {
        lds_put();                                            // group kt+1 -> LDS
        gload(kt + 2 < nk ? kt + 2 : nk - 1);
        lds_get(wfB, xfB);
        __builtin_amdgcn_sched_barrier(0);                  // keep the memory phase ahead of the MFMAs it overlaps with
        mma(wfA, xfA);
        __builtin_amdgcn_sched_barrier(0);
        lds_put();                                            // group kt+2 -> LDS
        gload(kt + 3 < nk ? kt + 3 : nk - 1);
        lds_get(wfA, xfA);
        __builtin_amdgcn_sched_barrier(0);
        mma(wfB, xfB);
        __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    // staging map: W tile n x 16 floats = n*4 quads -> 4 per thread (n = 512); X tile 80 x 16 = 320 quads -> threads < 320
    f32x4 wr[4], xr;
    const int xrow = tid >> 2, xq = tid & 3;                 // for tid < 320
    const bool xact = tid < SC_ROWS * 4;
    const bool xok = xact && xrow < N;
    auto load_tile = [&](int kt) {
        const int k0 = kt * SC_BK;
        const float* ab;
        int64_t lda;
        int kk;
        if (k0 < p.k1) { ab = p.a1; lda = p.lda1; kk = k0; }
        else           { ab = p.a2; lda = p.lda2; kk = k0 - p.k1; }
        if (xact) xr = *reinterpret_cast<const f32x4*>(ab + (row0 + (xok ? xrow : 0)) * lda + kk + xq * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + SC_T * i;
            const int r = f >> 2, q = f & 3;
            wr[i] = *reinterpret_cast<const f32x4*>(p.w + (int64_t)(r < n ? r : 0) * p.ldw + k0 + q * 4);
        }
    };
    auto store_tile = [&](float* stage) {
        float* Ws = stage;
        float* Xs = stage + SC_NMAX * SC_LD;
        if (xact) *reinterpret_cast<f32x4*>(Xs + xrow * SC_LD + xq * 4) = xok ? xr : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + SC_T * i;
            const int r = f >> 2, q = f & 3;
            *reinterpret_cast<f32x4*>(Ws + r * SC_LD + q * 4) = (r < n) ? wr[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto compute_tile = [&](const float* stage) {
        const float* Ws = stage + (wave * 64 + li) * SC_LD + lg * 4;
        const float* Xs = stage + SC_NMAX * SC_LD + li * SC_LD + lg * 4;
        f32x4 wf[4], xf[SC_TT];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) wf[ct] = *reinterpret_cast<const f32x4*>(Ws + ct * 16 * SC_LD);
#pragma unroll
        for (int tt = 0; tt < SC_TT; ++tt) xf[tt] = *reinterpret_cast<const f32x4*>(Xs + tt * 16 * SC_LD);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int tt = 0; tt < SC_TT; ++tt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    acc[tt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ct][s], xf[tt][s], acc[tt][ct], 0, 0, 0);
    };

    load_tile(0);
    store_tile(smem);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile(kt + 1);
        compute_tile(smem + (kt & 1) * SC_STAGE);
        if (kt + 1 < nk) store_tile(smem + ((kt + 1) & 1) * SC_STAGE);
        __syncthreads();
    }

  }
    // ---------------------------------------------------------------- epilogue (lane: token li + 16 tt, 4 channels)
    const int cbase = wave * 64 + lg * 4;                    // + 16 ct
    if (p.bias) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int c = cbase + 16 * ct;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float b = (c + e < n) ? p.bias[c + e] : 0.f;
#pragma unroll
                for (int tt = 0; tt < SC_TT; ++tt) acc[tt][ct][e] += b;
            }
        }
    }
    float mu = 0.f, rs = 1.f;
    if constexpr (GN) {
        // the wave holds the whole (scene, group): two-pass statistics over 64 channels x N tokens, wave-local
        const float inv_cnt = 1.0f / (64.0f * (float)N);
        float s = 0.f;
#pragma unroll
        for (int tt = 0; tt < SC_TT; ++tt)
            if (li + 16 * tt < N)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) s += (acc[tt][ct][0] + acc[tt][ct][1]) + (acc[tt][ct][2] + acc[tt][ct][3]);
        mu = wave_sum(s) * inv_cnt;
        s = 0.f;
#pragma unroll
        for (int tt = 0; tt < SC_TT; ++tt)
            if (li + 16 * tt < N)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = acc[tt][ct][e] - mu; s += d * d; }
        rs = 1.0f / sqrtf(wave_sum(s) * inv_cnt + p.eps);
    }
    const bool vec_ok = (p.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(p.y) & 15) == 0 && (n & 3) == 0;
    const bool res_vec = p.residual && (p.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(p.residual) & 15) == 0;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int c = cbase + 16 * ct;
        if (c >= n) continue;
        f32x4 ga = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
        if constexpr (GN) {
            ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
            be = *reinterpret_cast<const f32x4*>(p.beta + c);
        }
#pragma unroll
        for (int tt = 0; tt < SC_TT; ++tt) {
            const int tl = li + 16 * tt;
            if (tl >= N) continue;
            const int64_t tok = row0 + tl;
            f32x4 v = acc[tt][ct];
            if constexpr (GN) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (v[e] - mu) * rs * ga[e] + be[e];
                if (p.scale_shift) {
                    int64_t ssrow = tok;
                    if (p.ss_mode == DSC_SS_PER_SCENE) ssrow = scene;
                    else if (p.ss_mode == DSC_SS_PER_SLOT) ssrow = tl;
                    else if (p.ss_mode == DSC_SS_BY_INDEX) ssrow = p.ss_index[scene];
                    const float* ss = p.scale_shift + ssrow * p.ld_ss + c;
                    const f32x4 sc4 = *reinterpret_cast<const f32x4*>(ss);
                    const f32x4 sh4 = *reinterpret_cast<const f32x4*>(ss + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] * (sc4[e] + 1.0f) + sh4[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = dsc_silu_fast(v[e]);
            } else {
                if (p.act_out == DSC_ACT_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = dsc_gelu(v[e]);
                } else if (p.act_out == DSC_ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = dsc_silu(v[e]);
                }
            }
            if constexpr (MODE == 3) {
                // fragment-major activation tile (token tile tt, K group of the NEXT layer = this channel tile)
                const int64_t fo = (((int64_t)scene * SC_TT + tt) * (n / 16) + (wave * 4 + ct)) * 256 + lane * 4;
                if (p.residual) {
                    const f32x4 r4 = *reinterpret_cast<const f32x4*>(p.residual + fo);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += r4[e];
                }
                *reinterpret_cast<f32x4*>(p.y + fo) = v;
                continue;
            }
            if (p.residual) {
                if (res_vec && c + 3 < n) {
                    const f32x4 r4 = *reinterpret_cast<const f32x4*>(p.residual + tok * p.ldr + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += r4[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e < n) v[e] += p.residual[tok * p.ldr + c + e];
                }
            }
            if (vec_ok)
                *reinterpret_cast<f32x4*>(p.y + tok * p.ldy + c) = v;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < n) p.y[tok * p.ldy + c + e] = v[e];
            }
        }
    }
}

}  // namespace dsc_scene

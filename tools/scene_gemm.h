// Scene-resident layer kernel core (gfx950): ONE workgroup of 512 threads owns ONE scene (N <= 80 tokens) and computes a
// whole 1x1-conv / linear layer for it, D[channel][token] = sum_k W[channel][k] X[token][k] over n <= 512 output channels,
// so that consecutive layers of the denoiser can run inside one launch (scene_chain.hip): scenes are independent --
// GroupNorm, attention and LayerNorm never cross a scene -- hence no inter-block synchronisation is ever needed.
//
// Tiling: v_mfma_f32_16x16x4_f32 (80 tokens = 5 x 16, no padding).  Wave w owns channels [64w, 64w+64) = 4 channel tiles
// x 5 token tiles = 20 accumulator quads; for GroupNorm(8 groups of 64 channels) a wave therefore holds one complete
// (scene, group) and the statistics are wave-local (two-pass, no LDS, no barrier).
// Main loop: the block is split into two independent 4-wave groups (one wave per SIMD each); group g owns channels
// [256g, 256g+256), stages ITS weight rows and the scene's token rows for a 16-wide K group into its own double-buffered
// LDS tiles (coalesced loads shared by 4 waves: same staged bytes per MFMA as the tiled GEMM) and synchronises with an
// LDS-counter barrier of its own, so the groups drift out of phase and one group's staging / barrier / fragment-read gap
// is covered by the other group's MFMAs -- what two independent blocks per CU do for the tiled kernel.  LDS rows are
// padded to 24 floats: conflict-free ds_read_b128 for the 16x16x4 fragment pattern (lane (i = l%16, g = l/16) reads row
// i, floats [4g, 4g+4) = the k slots of 4 consecutive MFMAs; both operands use the same k permutation).
// Measured (tools/scene_tune.py, M=20480, n=512): 2.30-2.46 us per K group (MFMA-bound 2.2), launch 110.8 us plain /
// 112.6 us GN at K=512 vs 109.8 / 115.6 us for the tiled kernel.
#pragma once
#include "../diffuscene_amd/csrc/dsc_common.h"

namespace dsc_scene {

constexpr int SC_T = 512;              // threads per scene block
constexpr int SC_ROWS = 80;            // max tokens per scene
constexpr int SC_TT = 5;               // token tiles of 16
constexpr int SC_BK = 16;
constexpr int SC_LD = 24;              // padded LDS row
constexpr int SC_SMEM_FLOATS = 2 * 2 * (256 + SC_ROWS) * SC_LD + 64;   // two groups x two stages + barrier counters = 126 KB

// One layer for one scene.  `p` is the batch-level descriptor of the tiled GEMM (include/diffuscene_hip.h); rows
// [scene*N, scene*N + N) of every matrix belong to this block.  GN selects the fused GroupNorm+(scale,shift)+SiLU epilogue
// (p.n must be 512).  K = k1 + k2 must be a multiple of 32.  All 512 threads must call this; on return the block's stores
// are issued (the caller fences / synchronises before another layer reads them).
template <bool GN>
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

  {
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
  }
    // ---------------------------------------------------------------- epilogue (lane: token li + 16 tt, 4 channels)
    const int cbase = wave * 64 + lg * 4;                    // + 16 ct
    // residual quads are requested now, ahead of the statistics, into registers the main loop no longer needs
    const bool res_pre = p.residual && (p.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(p.residual) & 15) == 0 && n == 512;
    f32x4 rpre[SC_TT][4];
    if (res_pre) {
#pragma unroll
        for (int tt = 0; tt < SC_TT; ++tt) {
            const int tl = li + 16 * tt;
            const int64_t tok = row0 + (tl < N ? tl : 0);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
                rpre[tt][ct] = *reinterpret_cast<const f32x4*>(p.residual + tok * p.ldr + cbase + 16 * ct);
        }
    }
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
            if (res_pre) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += rpre[tt][ct][e];
            } else if (p.residual) {
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

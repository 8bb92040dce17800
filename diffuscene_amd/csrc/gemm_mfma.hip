// C ABI dispatch of the fp32 MFMA GEMM family (kernel template: gemm_core.h).
#include "gemm_core.h"
#include "gemm_skinny.h"
#include <cstdlib>

namespace {

using dsc_gemm::gemm_kernel;
constexpr int BK = 32;

int check_common(const dsc_gemm_args* a) {
    if (!a || !a->a1 || !a->w || !a->y) return DSC_EINVAL;
    if (a->m <= 0 || a->n <= 0 || a->k1 <= 0 || a->k2 < 0) return DSC_EINVAL;
    if ((a->k1 % BK) || (a->k2 % BK)) return DSC_EINVAL;
    if (a->k2 > 0 && !a->a2) return DSC_EINVAL;
    if (a->batch < 1) return DSC_EINVAL;
    if (!dsc_aligned16(a->a1) || (a->lda1 & 3) || !dsc_aligned16(a->w) || (a->ldw & 3)) return DSC_EALIGN;
    if (a->k2 > 0 && (!dsc_aligned16(a->a2) || (a->lda2 & 3))) return DSC_EALIGN;
    if (a->batch > 1 && ((a->sa1 & 3) || (a->sa2 & 3) || (a->sw & 3))) return DSC_EALIGN;
    if (a->act_in != DSC_ACT_NONE) return DSC_EINVAL;      // input activations are separate launches (dsc_activation_f32)
    if (a->act_out < DSC_ACT_NONE || a->act_out > DSC_ACT_SILU) return DSC_EINVAL;   // fused output activations: GELU / SiLU
    return 0;
}

// IL: interleaved main loop (gemm_core.h) -- every tile whose MFMA groups are long enough to carry the staging instructions.
// IL = 2 (operands DMA'd straight into LDS with fixed per-lane byte offsets) needs one row stride for both K segments and
// 32-bit byte offsets inside a tile; anything else takes the register-staged form (IL = 1), same arithmetic.
template <int TM, int TN, int WM, int WN, bool GN, int IL>
int launch_il(const dsc_gemm_args* a, int rows_per_blk, hipStream_t s) {
    constexpr int BN = 32 * TN * WN;
    const int nrb = (a->m + rows_per_blk - 1) / rows_per_blk;
    const int ncb = (a->n + BN - 1) / BN;
    dim3 grid((unsigned)(nrb * ncb), (unsigned)a->batch);
    // XCD-aware block order: the column blocks sharing a token tile run on one XCD and hit its L2; EPF: residual
    // quads are requested at the top of the epilogue (measured -3..-6 % per launch with a residual input)
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, GN, IL>), grid, dim3(64 * WM * WN), 0, s, *a, ncb);
    DSC_LAUNCH_CHECK();
    return 0;
}

template <int TM, int TN, int WM, int WN, bool GN, int IL = 1>
int launch(const dsc_gemm_args* a, int rows_per_blk, hipStream_t s) {
    if constexpr (IL == 0) {
        return launch_il<TM, TN, WM, WN, GN, 0>(a, rows_per_blk, s);
    } else {
        const int64_t ld_max = a->lda1 > a->ldw ? a->lda1 : a->ldw;
        const bool dma = (a->k2 == 0 || a->lda1 == a->lda2) && ld_max * 4 * (32 * TM * WM + 32 * TN * WN) < (int64_t(1) << 31);
        return dma ? launch_il<TM, TN, WM, WN, GN, 2>(a, rows_per_blk, s) : launch_il<TM, TN, WM, WN, GN, 1>(a, rows_per_blk, s);
    }
}

// out[r][c] = sum_s slabs[s][r][c] (+ bias[c]) (+ residual[r][c]); fixed summation order
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, long slab, int nslab, const float* __restrict__ bias,
                                     const float* __restrict__ residual, long ldr, float* __restrict__ out, long ldo, int m, int n) {
    const long total = (long)m * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / n), c = (int)(i - (long)r * n);
        float acc = 0.f;
        for (int s2 = 0; s2 < nslab; ++s2) acc += slabs[(long)s2 * slab + i];
        if (bias) acc += bias[c];
        if (residual) acc += residual[(long)r * ldr + c];
        out[(long)r * ldo + c] = acc;
    }
}

// ---- K-parallel kernel for launches too small to fill the chip (gemm_skinny.h) -------------------------------------------------
int g_skinny = -1;           // -1: read DSC_SKINNY on first use (default on)

int skinny_enabled() {
    if (g_skinny < 0) {
        const char* e = std::getenv("DSC_SKINNY");
        g_skinny = (e && e[0] == '0') ? 0 : 1;
    }
    return g_skinny;
}

struct SkinnyPlan { int rows_per_blk, ng, nrb, ncb; bool r16; };

// Does the K-parallel kernel take this launch?  Shape: K % 64 == 0 (eight slices of whole 8-k units), a K slice inside one operand
// segment, n % 64 == 0, whole scenes of <= 32 tokens per block, 16-byte aligned rows everywhere, and ALL blocks in one round of the
// chip -- beyond that the tile kernels' throughput wins over this kernel's latency (measured on the 128 x 12 text configuration with a
// 64-row form of this kernel: 21 us against the tile kernel's 19.5 us per GroupNorm launch; that form is gone).
bool skinny_plan(const dsc_gemm_args* a, bool gn, SkinnyPlan* out) {
    if (!skinny_enabled() || (a->flags & DSC_GEMM_ROW_INVARIANT)) return false;
    const int K = a->k1 + a->k2;
    if ((K % 64) || (a->n % 64)) return false;
    if (a->k2 > 0 && (a->k1 % (K / dsc_skinny::NW))) return false;
    if (a->preact && !gn) return false;
    if (a->actgrad_x || a->gnb_z) return false;
    int unit = 1;                                   // rows come in whole scenes under GroupNorm
    if (gn) {
        if (a->batch != 1) return false;
        unit = a->tokens_per_scene;
        if (unit > 32) return false;
    }
    const int ncb = a->n / 64;
    // blocks of <= 16 rows (16 x 16 MFMA tiles: half the MFMA time of a padded 32-row tile -- the 12-object scene) where the LDS-staged form
    // applies and they still fit one round; else blocks of <= 32 rows
    int rpb = 0;
    int64_t nrb = 0;
    bool r16 = false;
    for (int rows = ((K % 512) == 0 && unit <= 16) ? 16 : 32; rows <= 32 && !rpb; rows += 16) {
        int r = (rows / unit) * unit;
        if (r > a->m) r = (int)(((int64_t)a->m + unit - 1) / unit * unit);
        if (rows == 16 && r > 16) continue;
        const int64_t nb = ((int64_t)a->m + r - 1) / r;
        if (nb * ncb * a->batch <= 256) { rpb = r; nrb = nb; r16 = rows == 16; }
    }
    if (!rpb) return false;
    auto al = [](const void* q) { return !q || dsc_aligned16(q); };
    if (!al(a->y) || !al(a->residual) || !al(a->bias) || !al(a->preact) || !al(a->gamma) || !al(a->beta) || !al(a->scale_shift)) return false;
    if ((a->ldy & 3) || (a->residual && (a->ldr & 3)) || (a->preact && (a->ld_preact & 3)) || (a->scale_shift && (a->ld_ss & 3))) return false;
    if (a->batch > 1 && ((a->sy & 3) || (a->sres & 3) || (a->sbias & 3))) return false;
    out->rows_per_blk = rpb;
    out->r16 = r16;
    // K slices of whole 64-k chunks: the LDS-staged form; else fragments straight from global memory (register-resident up to K = 1024)
    out->ng = (K % 512) == 0 ? dsc_skinny::STAGED : K <= 256 ? 1 : K <= 512 ? 2 : K <= 1024 ? 4 : 0;
    out->nrb = (int)nrb;
    out->ncb = ncb;
    return true;
}

template <bool GN, int NG, bool R16 = false>
int launch_skinny_ng(const dsc_gemm_args* a, const SkinnyPlan& sp, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dsc_skinny::gemm_skinny_kernel<GN, NG, R16>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)dsc_skinny::lds_bytes<NG>());
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL((dsc_skinny::gemm_skinny_kernel<GN, NG, R16>), dim3((unsigned)(sp.nrb * sp.ncb), (unsigned)a->batch),
                       dim3(64 * dsc_skinny::NW), dsc_skinny::lds_bytes<NG>(), s, *a, sp.ncb, sp.rows_per_blk);
    DSC_LAUNCH_CHECK();
    return 0;
}

template <bool GN>
int launch_skinny(const dsc_gemm_args* a, const SkinnyPlan& sp, hipStream_t s) {
    if (sp.r16) return launch_skinny_ng<GN, dsc_skinny::STAGED, true>(a, sp, s);
    switch (sp.ng) {
        case 1: return launch_skinny_ng<GN, 1>(a, sp, s);
        case 2: return launch_skinny_ng<GN, 2>(a, sp, s);
        case 4: return launch_skinny_ng<GN, 4>(a, sp, s);
        case dsc_skinny::STAGED: return launch_skinny_ng<GN, dsc_skinny::STAGED>(a, sp, s);
        default: return launch_skinny_ng<GN, 0>(a, sp, s);
    }
}

}  // namespace

extern "C" int dsc_get_skinny(void) { return skinny_enabled(); }
extern "C" int dsc_set_skinny(int32_t on) { const int prev = skinny_enabled(); g_skinny = on ? 1 : 0; return prev; }
// 0: a tile kernel (or the split-bf16 family) takes the launch; 1: the K-parallel kernel with blocks of <= 32 rows, 2: of <= 16 rows (16 x 16 MFMA tiles)
extern "C" int dsc_gemm_skinny(const dsc_gemm_args* a, int32_t gn) {
    if (check_common(a) || dsc_gemm_arithmetic(a, gn)) return 0;
    SkinnyPlan sp;
    return skinny_plan(a, gn != 0, &sp) ? (sp.r16 ? 2 : 1) : 0;
}

// Tile choice: estimated time ~ ceil(blocks / 256 CUs) * tile area (every CU works through its blocks); ties go to the
// larger tile (fewer LDS-staged bytes per MFMA).
static long tile_cost(long m_rows, int rows_per_blk, int n, int bm, int bn) {
    const long nblk = ((m_rows + rows_per_blk - 1) / rows_per_blk) * ((n + bn - 1) / bn);
    return ((nblk + 255) / 256) * (long)bm * bn;
}

extern "C" int dsc_gemm_f32(const dsc_gemm_args* a, dsc_stream_t stream) {
    int rc = check_common(a);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = dsc_gemm_try_split(a, false, s);                // pre-split weight planes supplied: f32-accurate product on the bf16 pipe
    if (rc != DSC_SPLIT_NOT_TAKEN) return rc;
    if (a->preact || a->actgrad_x || a->gnb_z) return DSC_EINVAL;    // the training-step epilogues exist on the split kernels only: fail, never ignore
    {
        SkinnyPlan sp;
        if (skinny_plan(a, false, &sp)) return launch_skinny<false>(a, sp, s);
    }
    const bool wide = (a->n % 256) == 0;
    struct Cand { int bm, bn, id; };
    // ties go to the earlier candidate: 160 x 128 (2 blocks per CU) measured 1 % ahead of 160 x 256 at M = 20480
    const Cand cands[5] = {{160, 128, 1}, {160, 256, 0}, {128, 128, 2}, {96, 128, 3}, {64, 64, 4}};
    int best = -1;
    long best_cost = 0;
    for (int i = 0; i < 5; ++i) {
        if (cands[i].id == 0 && !wide) continue;
        const long c = tile_cost(a->m, cands[i].bm, a->n, cands[i].bm, cands[i].bn) * a->batch;
        if (best < 0 || c < best_cost) { best = cands[i].id; best_cost = c; }
    }
    switch (best) {
        // 8 waves x (5x1 tiles): 160 x 256 block tile, one block per CU (fewer LDS-staged bytes per MFMA, shortest
        // epilogue); needs full 256-column tiles
        case 0: return launch<5, 1, 1, 8, false>(a, 160, s);
        case 1: return launch<5, 1, 1, 4, false>(a, 160, s);
        case 2: return launch<2, 2, 2, 2, false>(a, 128, s);
        case 3: return launch<3, 1, 1, 4, false>(a, 96, s);
        default: return launch<1, 1, 2, 2, false, 0>(a, 64, s);
    }
}

extern "C" int dsc_gemm_gn_silu_f32(const dsc_gemm_args* a, dsc_stream_t stream) {
    int rc = check_common(a);
    if (rc) return rc;
    if (!a->gamma || !a->beta || !a->bias) return DSC_EINVAL;
    const int N = a->tokens_per_scene;
    if (N < 4 || N > 160) return DSC_ERANGE;
    if (a->m % N) return DSC_EINVAL;
    if (a->n % 128) return DSC_EINVAL;
    if (a->act_in != DSC_ACT_NONE) return DSC_EINVAL;
    if (!dsc_aligned16(a->gamma) || !dsc_aligned16(a->beta)) return DSC_EALIGN;
    if (a->preact && (!dsc_aligned16(a->preact) || (a->ld_preact & 3))) return DSC_EALIGN;
    if (a->ss_mode != DSC_SS_NONE) {
        if (!a->scale_shift) return DSC_EINVAL;
        if (!dsc_aligned16(a->scale_shift) || (a->ld_ss & 3)) return DSC_EALIGN;
        if (a->ss_mode < DSC_SS_NONE || a->ss_mode > DSC_SS_BY_INDEX) return DSC_EINVAL;
        if (a->ss_mode == DSC_SS_BY_INDEX && (!a->ss_index || a->ss_rows < 1)) return DSC_EINVAL;
    } else if (a->scale_shift) return DSC_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = dsc_gemm_try_split(a, true, s);
    if (rc != DSC_SPLIT_NOT_TAKEN) return rc;
    {
        SkinnyPlan sp;
        if (skinny_plan(a, true, &sp)) return launch_skinny<true>(a, sp, s);
    }
    // scene-aligned tiles: a block holds floor(BM / N) whole scenes; padded rows are wasted MFMA work
    struct Cand { int bm, bn; };
    const Cand cands[5] = {{160, 128}, {160, 256}, {128, 128}, {96, 128}, {64, 64}};
    int best = -1;
    long best_cost = 0;
    // (scale, shift) rows addressed per scene (PER_SCENE, BY_INDEX) are staged in LDS at kernel start: the tile must not hold
    // more scenes than the kernel has slots for (gemm_core.h SSL_MAX: 4 for the 4-wave 160 x 128 tile, 8 otherwise)
    const bool per_scene_ss = a->ss_mode == DSC_SS_PER_SCENE || a->ss_mode == DSC_SS_BY_INDEX;
    for (int pass = 0; pass < 2 && best < 0; ++pass)
        for (int i = 0; i < 5; ++i) {
            if (cands[i].bm < N) continue;
            if (cands[i].bn == 256 && (a->n % 256)) continue;
            if (pass == 0 && per_scene_ss && cands[i].bm / N > ((cands[i].bm == 160 && cands[i].bn == 128) ? 4 : 8)) continue;
            const int rpb = (cands[i].bm / N) * N;
            const long c = tile_cost(a->m, rpb, a->n, cands[i].bm, cands[i].bn);
            if (best < 0 || c < best_cost) { best = i; best_cost = c; }
        }
    if (best < 0) return DSC_ERANGE;
    const int rpb = (cands[best].bm / N) * N;
    switch (best) {
        case 0: return launch<5, 1, 1, 4, true>(a, rpb, s);
        case 1: return launch<5, 1, 1, 8, true>(a, rpb, s);
        case 2: return launch<2, 2, 2, 2, true>(a, rpb, s);
        case 3: return launch<3, 1, 1, 4, true>(a, rpb, s);
        default: return launch<1, 1, 2, 2, true, 0>(a, rpb, s);
    }
}

// Split-K form for short, deep products (m <= 512 rows, K >= 1024: the time / context MLP layers of the training step, where
// an output-tile-parallel launch has 4..128 blocks each walking K serially -- 52 us for a 2 GFLOP product).  The K range is cut
// into `splits` equal parts that run as the batch dimension of ONE dsc_gemm_f32 launch into workspace slabs [splits][m][n];
// a second kernel sums the slabs in a fixed order and applies bias / residual.  No output activation, single K segment.
extern "C" int dsc_gemm_splitk_f32(const dsc_gemm_args* a, int32_t splits, float* workspace, int64_t workspace_floats,
                                   dsc_stream_t stream) {
    int rc = check_common(a);
    if (rc) return rc;
    if (splits < 2 || splits > 64 || !workspace) return DSC_EINVAL;
    if (a->batch != 1 || a->k2 != 0 || a->act_out != DSC_ACT_NONE) return DSC_EINVAL;
    if (a->k1 % (BK * splits)) return DSC_EINVAL;
    if (workspace_floats < (int64_t)splits * a->m * a->n) return DSC_EINVAL;
    if ((reinterpret_cast<uintptr_t>(workspace) & 15) || (((int64_t)a->m * a->n) & 3)) return DSC_EALIGN;
    dsc_gemm_args p = *a;
    const int kc = a->k1 / splits;
    p.k1 = kc;
    p.batch = splits;
    p.sa1 = kc; p.sw = kc; p.sa2 = 0; p.sbias = 0; p.sres = 0;
    p.bias = nullptr; p.residual = nullptr;
    p.y = workspace; p.ldy = a->n; p.sy = (int64_t)a->m * a->n;
    rc = dsc_gemm_f32(&p, stream);
    if (rc) return rc;
    const long total = (long)a->m * a->n;
    long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), workspace,
                       (long)a->m * a->n, splits, a->bias, a->residual, (long)a->ldr, a->y, (long)a->ldy, a->m, a->n);
    DSC_LAUNCH_CHECK();
    return 0;
}

// y = LayerNorm_channels([a | a2] @ W^T + bias) * gain (+ residual): the out-projection + LayerNorm (+ PreNorm residual) of
// LinearAttention / LinearAttentionCross (denoise_net.py:216-235, :98-102) in one launch.  Blocks of 96 token rows span all
// n = 512 channels (8 waves x (3 x 2) tiles, LDS-DMA interleaved main loop), so the per-token statistics stay inside the block.
extern "C" int dsc_gemm_layernorm_f32(const dsc_gemm_args* a, dsc_stream_t stream) {
    int rc = check_common(a);
    if (rc) return rc;
    if (a->n != 512 || a->batch != 1 || a->act_out != DSC_ACT_NONE || !a->gamma) return DSC_EINVAL;
    if (!dsc_aligned16(a->gamma) || !dsc_aligned16(a->y) || (a->ldy & 3)) return DSC_EALIGN;
    if (a->residual && (!dsc_aligned16(a->residual) || (a->ldr & 3))) return DSC_EALIGN;
    if (a->bias && !dsc_aligned16(a->bias)) return DSC_EALIGN;
    const int64_t ld_max = a->lda1 > a->ldw ? a->lda1 : a->ldw;
    if (!(a->k2 == 0 || a->lda1 == a->lda2) || ld_max * 4 * (96 + 512) >= (int64_t(1) << 31)) return DSC_ERANGE;
    const int nrb = (a->m + 95) / 96;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL((gemm_kernel<3, 2, 1, 8, false, 2, 0, true>), dim3((unsigned)nrb, 1), dim3(512), 0,
                       static_cast<hipStream_t>(stream), *a, 1);
    DSC_LAUNCH_CHECK();
    return 0;
}

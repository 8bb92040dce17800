// C ABI dispatch of the fp32 MFMA GEMM family (kernel template: gemm_core.h).
#include "gemm_core.h"

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
    if (a->act_in != DSC_ACT_NONE && a->act_in != DSC_ACT_SILU) return DSC_EINVAL;
    return 0;
}

template <int TM, int TN, int WM, int WN, bool GN, int BKT = 32>
int launch(const dsc_gemm_args* a, int rows_per_blk, hipStream_t s) {
    constexpr int BN = 32 * TN * WN;
    const int nrb = (a->m + rows_per_blk - 1) / rows_per_blk;
    const int ncb = (a->n + BN - 1) / BN;
    dim3 grid((unsigned)(nrb * ncb), (unsigned)a->batch);
    // XCD-aware block order: the column blocks sharing a token tile run on one XCD and hit its L2
    hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, GN, BKT, false, 2, true>), grid, dim3(64 * WM * WN), 0, s, *a, ncb);
    DSC_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int dsc_gemm_f32(const dsc_gemm_args* a, dsc_stream_t stream) {
    int rc = check_common(a);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // pick the token-tile height that wastes the fewest padded rows (ties -> taller tile)
    const int cand[3] = {160, 128, 96};
    int best = 160;
    long best_cost = -1;
    for (int i = 0; i < 3; ++i) {
        const long cost = (long)((a->m + cand[i] - 1) / cand[i]) * cand[i];
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = cand[i]; }
    }
    if (best == 160) {
        // 8 waves x (5x1 tiles): 160 x 256 block tile, one block per CU -- fewer LDS-staged bytes per MFMA (measured
        // +10 % over the 4-wave 160 x 128 tile on M=20480, n=512); needs full 256-column tiles and K tiles of 64
        if ((a->n % 256) == 0 && (a->k1 % 64) == 0 && (a->k2 % 64) == 0) return launch<5, 1, 1, 8, false, 64>(a, 160, s);
        return launch<5, 1, 1, 4, false>(a, 160, s);
    }
    if (best == 128) return launch<2, 2, 2, 2, false>(a, 128, s);
    return launch<3, 1, 1, 4, false>(a, 96, s);
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
        if (a->ss_mode < DSC_SS_NONE || a->ss_mode > DSC_SS_PER_SLOT) return DSC_EINVAL;
    } else if (a->scale_shift) return DSC_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int cand[3] = {160, 128, 96};
    int best = 0;
    double best_u = -1.0;
    for (int i = 0; i < 3; ++i) {
        if (cand[i] < N) continue;
        const double u = (double)((cand[i] / N) * N) / cand[i];
        if (u > best_u + 1e-9) { best_u = u; best = cand[i]; }
    }
    if (!best) return DSC_ERANGE;
    const int rpb = (best / N) * N;
    if (best == 160) return launch<5, 1, 1, 4, true>(a, rpb, s);
    if (best == 128) return launch<2, 2, 2, 2, true>(a, rpb, s);
    return launch<3, 1, 1, 4, true>(a, rpb, s);
}

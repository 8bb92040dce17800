// C ABI entry of the grouped weight-gradient GEMM on the split-bf16 path (kernel body: gemm_tn_split.h).  Own translation unit: built
// with -fno-slp-vectorize like gemm_split.hip (scalar f32 subtractions in the operand split: packed f32 VALU beside MFMAs costs
// issue slots), which the other backward kernels of train.hip should not inherit.
#include "gemm_tn_split.h"
#include <atomic>
#include <cstdlib>
#include <cstring>

#ifndef DSC_TN_NCW
#define DSC_TN_NCW 4          // consumer waves of the producer / consumer form: 4 (64 x 128 wave tiles, 512 threads: product) or 8 (64 x 64, 768 threads)
#endif
#ifndef DSC_TN_WS
#define DSC_TN_WS 1          // 1: producer / consumer waves (12 waves per block, round 5); 0: the round-4 block (same-box A/B builds)
#endif

// The same grouped weight-gradient launch on the bf16 matrix cores (gemm_tn_split.h: operands split exactly into three bf16 pieces,
// six products, f32 accumulation -- error vs f64 <= the f32-MFMA kernel's, ~1.6x faster).  Tiles are 256 (n) x 128 (k), addressed
// through the host's block map (below); the slab reduction is the f32 form's (128 x 128 tiles, tile0).
// Block placement (round 4).  The hardware deals consecutive workgroup ids round-robin over the 8 XCDs, each with its own 4 MB L2, and the
// 256 x 128 tiles of one layer share its operand strips (a 512 x 512 dW: 8 tiles, every dY strip read by 4 of them, every A strip by 2).
// With tiles numbered layer by layer a layer's tiles sat on 8 DIFFERENT L2s and every strip crossed the fabric once per tile (measured
// round 3: 9.1 GB fetched per launch, 3x the operands).  The host now hands the kernel a block map -- physical block id -> (group, tile
// of the group) -- that puts the tiles of one layer on ONE XCD (ids congruent mod 8), long layers first on every XCD, the short groups'
// tiles dealt out evenly (train_plan.tn_block_map); (-1, -1) pads the XCD lists to one length.  tools/tn_probe.py, same step, 256
// blocks: 1632 -> 1564 us (-4.2 %; all strips hot in every L2: 1520).
__global__ __launch_bounds__(512, 1) void gemm_tn_split_grouped_kernel(const dsc_tn_group* __restrict__ groups, const int2* __restrict__ block_map,
                                                                       const int splits, float* __restrict__ workspace) {
    __shared__ __attribute__((aligned(16))) char smem[dsc_tn_split::SMEM];
    const int2 gt = block_map[blockIdx.x];
    if (gt.x < 0) return;
    const dsc_tn_group g = groups[gt.x];
    const int K = g.k1 + g.k2;
    const int ktiles = (K + 127) / 128;
    const int local = gt.y;
    const int split = blockIdx.y;
    dsc_tn_split::Prob p;
    p.a1 = g.a1; p.lda1 = g.lda1; p.k1 = g.k1; p.a2 = g.a2; p.lda2 = g.lda2; p.k2 = g.k2; p.dy = g.dy; p.ldd = g.ldd;
    p.m = g.m; p.n = g.n; p.kvalid = g.kvalid;
    p.chunk = ((g.m + splits - 1) / splits + 31) / 32 * 32;
    if (splits == 1) {
        p.out = g.out; p.ldo = g.ldo; p.bias_out = g.dbias; p.slab = 0; p.bias_slab = 0;
    } else {
        const long wslab = (long)g.n * g.kvalid;
        p.out = workspace + g.ws_offset; p.ldo = g.kvalid; p.slab = wslab;
        p.bias_out = g.dbias ? workspace + g.ws_offset + wslab * splits : nullptr; p.bias_slab = g.n;
    }
    dsc_tn_split::tn_split_block<0>(p, local % ktiles, local / ktiles, split, smem);
}

// The same launch with producer and consumer waves (gemm_tn_split.h, tn_split_block_ws): 768 threads, identical results.
constexpr int DSC_TN_WS_THREADS = (DSC_TN_NCW + dsc_tn_split::NPW) * 64;
__global__ __launch_bounds__(DSC_TN_WS_THREADS, 1) void gemm_tn_split_grouped_ws_kernel(const dsc_tn_group* __restrict__ groups,
                                                                                         const int2* __restrict__ block_map, const int splits,
                                                                                         float* __restrict__ workspace) {
    __shared__ __attribute__((aligned(16))) char smem[dsc_tn_split::SMEM];
    const int2 gt = block_map[blockIdx.x];
    if (gt.x < 0) return;
    const dsc_tn_group g = groups[gt.x];
    const int K = g.k1 + g.k2;
    const int ktiles = (K + 127) / 128;
    const int local = gt.y;
    const int split = blockIdx.y;
    dsc_tn_split::Prob p;
    p.a1 = g.a1; p.lda1 = g.lda1; p.k1 = g.k1; p.a2 = g.a2; p.lda2 = g.lda2; p.k2 = g.k2; p.dy = g.dy; p.ldd = g.ldd;
    p.m = g.m; p.n = g.n; p.kvalid = g.kvalid;
    p.chunk = ((g.m + splits - 1) / splits + 31) / 32 * 32;
    if (splits == 1) {
        p.out = g.out; p.ldo = g.ldo; p.bias_out = g.dbias; p.slab = 0; p.bias_slab = 0;
    } else {
        const long wslab = (long)g.n * g.kvalid;
        p.out = workspace + g.ws_offset; p.ldo = g.kvalid; p.slab = wslab;
        p.bias_out = g.dbias ? workspace + g.ws_offset + wslab * splits : nullptr; p.bias_slab = g.n;
    }
    dsc_tn_split::tn_split_block_ws<DSC_TN_NCW>(p, local % ktiles, local / ktiles, split, smem);
}

// Round 6: 256 x 256 tiles, four waves with the 512-register budget (gemm_tn_split.h, tn_split_block_big); the block map addresses
// 256-wide k tiles (tile_k = 256 in the launch).
__global__ __launch_bounds__(dsc_tn_split::BT, 1) void gemm_tn_split_grouped_big_kernel(const dsc_tn_group* __restrict__ groups,
                                                                                         const int2* __restrict__ block_map, const int splits,
                                                                                         float* __restrict__ workspace) {
    __shared__ __attribute__((aligned(16))) char smem[dsc_tn_split::BSMEM];
    const int2 gt = block_map[blockIdx.x];
    if (gt.x < 0) return;
    const dsc_tn_group g = groups[gt.x];
    const int K = g.k1 + g.k2;
    const int ktiles = (K + 255) / 256;
    const int local = gt.y;
    const int split = blockIdx.y;
    dsc_tn_split::Prob p;
    p.a1 = g.a1; p.lda1 = g.lda1; p.k1 = g.k1; p.a2 = g.a2; p.lda2 = g.lda2; p.k2 = g.k2; p.dy = g.dy; p.ldd = g.ldd;
    p.m = g.m; p.n = g.n; p.kvalid = g.kvalid;
    p.chunk = ((g.m + splits - 1) / splits + 31) / 32 * 32;
    if (splits == 1) {
        p.out = g.out; p.ldo = g.ldo; p.bias_out = g.dbias; p.slab = 0; p.bias_slab = 0;
    } else {
        const long wslab = (long)g.n * g.kvalid;
        p.out = workspace + g.ws_offset; p.ldo = g.kvalid; p.slab = wslab;
        p.bias_out = g.dbias ? workspace + g.ws_offset + wslab * splits : nullptr; p.bias_slab = g.n;
    }
    dsc_tn_split::tn_split_block_big(p, local % ktiles, local / ktiles, split, smem);
}

// Which block body the split-bf16 weight-gradient launch runs: 2 = 256 x 256 tiles, four waves with the 512-register budget (round 6,
// default); 1 = 256 x 128 tiles with producer / consumer waves (round 5); 0 = the round-4 block (every wave stages and multiplies).
// Identical results; the switch exists so that a test can hold the three to each other bit for bit.  The tile width is baked into the
// host's block map, so the LAUNCH says which numbering its map uses (tile_k); the form picks the body among those of that width.
// Initial value: DSC_TN_FORM in the environment ("", "2" -> 2; "1"; "0"; anything else -> 2 is NOT assumed: the launch fails).
static std::atomic<int> g_tn_form{-1000};
static int tn_form() {
    int f = g_tn_form.load(std::memory_order_relaxed);
    if (f == -1000) {
        const char* e = getenv("DSC_TN_FORM");
        f = (!e || !e[0] || !strcmp(e, "2")) ? 2 : !strcmp(e, "1") ? 1 : !strcmp(e, "0") ? 0 : DSC_EINVAL;
        g_tn_form.store(f, std::memory_order_relaxed);
    }
    return f;
}
extern "C" int dsc_get_tn_split_form(void) { return tn_form(); }
extern "C" int dsc_set_tn_split_form(int32_t form) {
    if (form < 0 || form > 2) return DSC_EINVAL;
    const int prev = tn_form();
    g_tn_form.store(form, std::memory_order_relaxed);
    return prev;
}

extern "C" int dsc_gemm_tn_grouped_split_f32(const dsc_tn_group* groups_dev, int32_t count, int32_t total_tiles,
                                             const int32_t* block_map_dev, int32_t blocks, int32_t splits, float* workspace,
                                             int64_t workspace_floats, int64_t workspace_needed, int32_t tile_k, dsc_stream_t stream) {
    if (!groups_dev || !block_map_dev || count < 1 || total_tiles < count || blocks < count || splits < 1 || splits > 64) return DSC_EINVAL;
    if (tile_k != 128 && tile_k != 256) return DSC_EINVAL;
    if (reinterpret_cast<uintptr_t>(block_map_dev) & 7) return DSC_EALIGN;
    if (splits > 1 && (!workspace || workspace_floats < workspace_needed || workspace_needed < 1)) return DSC_EINVAL;
    const int form = tn_form();
    if (form < 0) return DSC_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    DSC_CLEAR_STALE_ERROR();
    if (tile_k == 256)
        hipLaunchKernelGGL(gemm_tn_split_grouped_big_kernel, dim3((unsigned)blocks, (unsigned)splits), dim3(dsc_tn_split::BT), 0, s, groups_dev,
                           reinterpret_cast<const int2*>(block_map_dev), splits, workspace);
    else if (form != 0)
        hipLaunchKernelGGL(gemm_tn_split_grouped_ws_kernel, dim3((unsigned)blocks, (unsigned)splits), dim3(DSC_TN_WS_THREADS), 0, s, groups_dev,
                           reinterpret_cast<const int2*>(block_map_dev), splits, workspace);
    else
        hipLaunchKernelGGL(gemm_tn_split_grouped_kernel, dim3((unsigned)blocks, (unsigned)splits), dim3(512), 0, s, groups_dev,
                           reinterpret_cast<const int2*>(block_map_dev), splits, workspace);
    DSC_LAUNCH_CHECK();
    if (splits > 1) return dsc_launch_reduce_grouped(groups_dev, count, total_tiles, splits, workspace, s);
    return 0;
}

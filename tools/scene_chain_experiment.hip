// Scene-resident chain of layers: up to DSC_CHAIN_MAX consecutive 1x1-conv / linear layers of the
// denoiser -- e.g. a whole ResnetBlock: WS-conv+GN+SiLU, res_conv, WS-conv+GN+SiLU+residual (denoise_net.py:178-206) --
// executed by ONE launch in which every workgroup owns one scene and walks the layer list (scene_gemm.h).  Between two
// layers only the block itself synchronises: no kernel boundary, no launch gap, GroupNorm statistics are wave-local, and
// the intermediate activations are re-read by the CU that just wrote them.  Any run of consecutive GEMM steps of a launch
// plan may be chained, whatever their data flow: layer i+1 starts after every wave of the block finished layer i, and
// layers never read another scene's rows.
#include "scene_gemm.h"   // tools/: round-1/2 experiment, not built into the library

namespace {

struct ChainArgs {
    dsc_gemm_args op[DSC_CHAIN_MAX];
    int gn[DSC_CHAIN_MAX];
    int count;
    int n_tok;
};

__global__ __launch_bounds__(512, 1) void scene_chain_kernel(const ChainArgs c) {
    extern __shared__ __attribute__((aligned(16))) float scene_smem[];
    const int scene = blockIdx.x;
    for (int i = 0; i < c.count; ++i) {
        if (c.gn[i]) dsc_scene::scene_gemm<true>(c.op[i], scene, c.n_tok, scene_smem);
        else dsc_scene::scene_gemm<false>(c.op[i], scene, c.n_tok, scene_smem);
        __syncthreads();       // workgroup-scope release/acquire: this layer's stores are visible to every wave of the block
    }
}

int check_op(const dsc_gemm_args& a, int gn, int n_tok, int64_t m) {
    if (!a.a1 || !a.w || !a.y) return DSC_EINVAL;
    if (a.m != m || a.n <= 0 || a.n > 512 || a.k1 <= 0 || a.k2 < 0) return DSC_EINVAL;
    if ((a.k1 % 32) || (a.k2 % 32)) return DSC_EINVAL;
    if (a.k2 > 0 && !a.a2) return DSC_EINVAL;
    if (a.batch != 1 || a.act_in != DSC_ACT_NONE) return DSC_EINVAL;
    if (!dsc_aligned16(a.a1) || (a.lda1 & 3) || !dsc_aligned16(a.w) || (a.ldw & 3)) return DSC_EALIGN;
    if (a.k2 > 0 && (!dsc_aligned16(a.a2) || (a.lda2 & 3))) return DSC_EALIGN;
    if (a.preact) return DSC_EINVAL;                     // the training forward keeps the tiled kernel
    if (gn) {
        if (a.n != 512 || !a.gamma || !a.beta || !a.bias) return DSC_EINVAL;
        if (a.tokens_per_scene != n_tok) return DSC_EINVAL;
        if (!dsc_aligned16(a.gamma) || !dsc_aligned16(a.beta)) return DSC_EALIGN;
        if ((a.ldy & 3) || !dsc_aligned16(a.y)) return DSC_EALIGN;
        if (a.ss_mode != DSC_SS_NONE) {
            if (!a.scale_shift || a.ss_mode < DSC_SS_NONE || a.ss_mode > DSC_SS_BY_INDEX) return DSC_EINVAL;
            if (!dsc_aligned16(a.scale_shift) || (a.ld_ss & 3)) return DSC_EALIGN;
            if (a.ss_mode == DSC_SS_BY_INDEX && !a.ss_index) return DSC_EINVAL;
        } else if (a.scale_shift) return DSC_EINVAL;
    } else if (a.gamma || a.beta || a.scale_shift) return DSC_EINVAL;
    return 0;
}

}  // namespace

extern "C" int dsc_scene_chain_f32(const dsc_gemm_args* ops, const int32_t* is_gn, int32_t count, int32_t tokens_per_scene,
                                   dsc_stream_t stream) {
    if (!ops || !is_gn || count < 1 || count > DSC_CHAIN_MAX) return DSC_EINVAL;
    if (tokens_per_scene < 1 || tokens_per_scene > dsc_scene::SC_ROWS) return DSC_ERANGE;
    const int64_t m = ops[0].m;
    if (m <= 0 || (m % tokens_per_scene)) return DSC_EINVAL;
    ChainArgs c;
    for (int i = 0; i < count; ++i) {
        const int rc = check_op(ops[i], is_gn[i], tokens_per_scene, m);
        if (rc) return rc;
        c.op[i] = ops[i];
        c.gn[i] = is_gn[i] ? 1 : 0;
    }
    c.count = count;
    c.n_tok = tokens_per_scene;
    const size_t lds = dsc_scene::SC_SMEM_FLOATS * sizeof(float);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(scene_chain_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(scene_chain_kernel, dim3((unsigned)(m / tokens_per_scene)), dim3(512), lds,
                       static_cast<hipStream_t>(stream), c);
    DSC_LAUNCH_CHECK();
    return 0;
}

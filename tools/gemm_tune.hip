// Tuning harness (NOT part of the product library): instantiates variants of the product GEMM kernel template
// (diffuscene_amd/csrc/gemm_core.h) so tools/gemm_tune.py can time them on a real MI355X.
#define DSC_GEMM_TIMING 1
#include "../diffuscene_amd/csrc/gemm_core.h"
#include "gemm_core_r01.h"

__device__ long long g_dsc_timing[4096 * 8];
__device__ int g_dsc_stagger = 0;
extern "C" int tune_set_stagger(int cycles) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dsc_stagger), &cycles, sizeof(int)); }

extern "C" int tune_read_timing(long long* host, int nblocks) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dsc_timing), sizeof(long long) * 8 * nblocks);
}

using dsc_gemm::gemm_kernel;

template <int TM, int TN, int WM, int WN, bool GN, int IL, int PROBE = 0>
static int run(const dsc_gemm_args* a, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    const int rpb = GN ? (BM / a->tokens_per_scene) * a->tokens_per_scene : BM;
    const int nrb = (a->m + rpb - 1) / rpb, ncb = (a->n + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, GN, IL, PROBE>), dim3(nrb * ncb, a->batch), dim3(64 * WM * WN), 0, s, *a, ncb);
    return (int)hipGetLastError();
}

template <int TM, int TN, int WM, int WN, bool GN, int BK>
static int run_r01(const dsc_gemm_args* a, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    const int rpb = GN ? (BM / a->tokens_per_scene) * a->tokens_per_scene : BM;
    const int nrb = (a->m + rpb - 1) / rpb, ncb = (a->n + BN - 1) / BN;
    hipLaunchKernelGGL((dsc_gemm_r01::gemm_kernel<TM, TN, WM, WN, GN, BK, false, 2, true, false, true>), dim3(nrb * ncb, a->batch),
                       dim3(64 * WM * WN), 0, s, *a, ncb);
    return (int)hipGetLastError();
}

#define VI(id, TM, TN, WM, WN, IL) \
    case id: return gn ? run<TM, TN, WM, WN, true, IL>(a, s) : run<TM, TN, WM, WN, false, IL>(a, s);

// Variant ids are kept from the round-1/2 logs under profiles/; the double-buffered / software-pipelined template branches that
// ids 0-22 and 24 instantiated were removed from the product header after the interleaved loops superseded them (their
// measurements: profiles/r01_gemm_tune_variants.txt, profiles/r02_gemm_ab_round1_variants_warm.txt).
extern "C" int tune_launch2(int variant, int gn, const dsc_gemm_args* a, void* stream, int stagger);
extern "C" int tune_launch(int variant, int gn, const dsc_gemm_args* a, void* stream) { return tune_launch2(variant, gn, a, stream, 0); }
extern "C" int tune_launch2(int variant, int gn, const dsc_gemm_args* a, void* stream, int) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (variant) {
        VI(23, 5, 1, 1, 4, 0)
        case 25: return gn ? run_r01<5, 1, 1, 4, true, 32>(a, s) : run_r01<5, 1, 1, 4, false, 32>(a, s);
        case 26: return gn ? -1 : run_r01<5, 1, 1, 8, false, 64>(a, s);
        case 30: return run<5, 1, 1, 4, false, 0, 1>(a, s);
        case 31: return run<5, 1, 1, 4, false, 0, 2>(a, s);
        case 32: return run<5, 1, 1, 4, false, 0, 3>(a, s);
        VI(33, 5, 1, 1, 8, 1)
        VI(34, 5, 1, 1, 4, 1)
        case 35: return run<5, 1, 1, 8, false, 1, 1>(a, s);
        case 36: return run<5, 1, 1, 8, false, 1, 2>(a, s);
        VI(37, 5, 1, 1, 8, 2)
        VI(38, 5, 1, 1, 4, 2)
        VI(39, 3, 1, 1, 4, 2)
        VI(40, 2, 2, 2, 2, 2)
    }
    return -1;
}

extern "C" const char* tune_name(int variant) {
    static const char* names[] = {
        "0: 160x128 4w BK32 single 2blk/CU (round-1 baseline)", "1: baseline + XCD remap",
        "2: 160x128 4w BK16 double-buffered", "3: 160x128 4w BK16 DB + XCD",
        "4: 160x256 8w BK32 DB", "5: 160x256 8w BK32 DB + XCD", "6: 160x256 8w BK32 single + XCD",
        "7: 160x256 8w BK64 single + XCD", "8: 160x256 4w(5x2 tiles) BK32 DB 1 wave/SIMD + XCD",
        "9: 160x128 4w BK64 single + XCD", "10: 160x128 4w BK32 DB (1 blk/CU) + XCD",
        "11: 160x256 8w BK16 DB + XCD", "12: 160x256 4w(5x2) BK16 DB + XCD",
        "13: PIPE 160x128 4w BK16 (2 blk/CU)", "14: PIPE 160x256 8w BK32 (1 blk/CU)", "15: PIPE 160x256 8w BK16",
        "16: PIPE 160x128 4w BK32 (1 blk/CU)", "17: PIPE 160x256 4w(5x2) BK16 1 wave/SIMD",
        "18: 64x64 4w (many small blocks)", "19: 128x128 4w 3 waves/SIMD", "20: 128x64 4w", "21: 96x128 4w", "22: 64x128 4w",
        "23: classic loop 160x128 4w (round-1 structure, round-2 epilogue)", "24: product 160x256 8w BK64 + epilogue residual prefetch",
        "25: ROUND-1 product 160x128 4w (git a07da01)", "26: ROUND-1 product 160x256 8w BK64 (git a07da01)",
        "27: PROBE 8w BK64: no global loads / LDS stores", "28: PROBE 8w BK64: + fragments read once", "29: PROBE 8w BK64: + no barriers",
        "30: PROBE 4w BK32: no global loads / LDS stores", "31: PROBE 4w BK32: + fragments read once", "32: PROBE 4w BK32: + no barriers",
        "33: IL 160x256 8w BK32 interleaved staging (1 blk/CU)", "34: IL 160x128 4w BK32 interleaved staging (2 blk/CU)",
        "35: PROBE IL 8w: no staging (wrong results)", "36: PROBE IL 8w: no staging, no barrier",
        "37: IL2 160x256 8w LDS-DMA staging (1 blk/CU)", "38: IL2 160x128 4w LDS-DMA staging (2 blk/CU)",
        "39: IL2 96x128 4w LDS-DMA", "40: IL2 128x128 4w (2x2 tiles) LDS-DMA"};
    return (variant >= 0 && variant < 41) ? names[variant] : nullptr;
}

// ---- scene-resident layer kernel (diffuscene_amd/csrc/scene_core.h): one block of 512 threads per scene -------------
#include "scene_core.h"

template <bool GN>
__global__ __launch_bounds__(512, 1) void scene_layer_kernel(const dsc_gemm_args p) {
    extern __shared__ __attribute__((aligned(16))) float scene_smem[];
    dsc_scene::scene_gemm<GN, 0>(p, blockIdx.x, p.tokens_per_scene, scene_smem);
}

template <bool GN>
__global__ __launch_bounds__(512, 1) void scene_layer_direct_kernel(const dsc_gemm_args p) {
    extern __shared__ __attribute__((aligned(16))) float scene_smem[];
    dsc_scene::scene_gemm<GN, 2>(p, blockIdx.x, p.tokens_per_scene, scene_smem);
}

template <int PROBE>
__global__ __launch_bounds__(512, 1) void scene_layer_probe_kernel(const dsc_gemm_args p) {
    extern __shared__ __attribute__((aligned(16))) float scene_smem[];
    dsc_scene::scene_gemm<true, 2, PROBE>(p, blockIdx.x, p.tokens_per_scene, scene_smem);
}
extern "C" int tune_scene_probe_launch(int probe, const dsc_gemm_args* a, void* stream) {
    const int scenes = a->m / a->tokens_per_scene;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = dsc_scene::SC_WP_SMEM_FLOATS * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scene_layer_probe_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scene_layer_probe_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    if (probe == 1) hipLaunchKernelGGL(scene_layer_probe_kernel<1>, dim3(scenes), dim3(512), lds, s, *a);
    else hipLaunchKernelGGL(scene_layer_probe_kernel<2>, dim3(scenes), dim3(512), lds, s, *a);
    return (int)hipGetLastError();
}

template <bool GN>
__global__ __launch_bounds__(512, 1) void scene_layer_g2_kernel(const dsc_gemm_args p) {
    extern __shared__ __attribute__((aligned(16))) float scene_smem[];
    dsc_scene::scene_gemm<GN, 4>(p, blockIdx.x, p.tokens_per_scene, scene_smem);
}
extern "C" int tune_scene_g2_launch(int gn, const dsc_gemm_args* a, void* stream) {
    const size_t lds = dsc_scene::SC_G2_SMEM_FLOATS * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scene_layer_g2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scene_layer_g2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int scenes = a->m / a->tokens_per_scene;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (gn) hipLaunchKernelGGL(scene_layer_g2_kernel<true>, dim3(scenes), dim3(512), lds, s, *a);
    else hipLaunchKernelGGL(scene_layer_g2_kernel<false>, dim3(scenes), dim3(512), lds, s, *a);
    return (int)hipGetLastError();
}

template <bool GN, int PROBE>
__global__ __launch_bounds__(512, 1) void scene_layer_frag_kernel(const dsc_gemm_args p) {
    dsc_scene::scene_gemm<GN, 3, PROBE>(p, blockIdx.x, p.tokens_per_scene, nullptr);
}
extern "C" int tune_scene_frag_launch(int gn, int probe, const dsc_gemm_args* a, void* stream) {
    const int scenes = a->m / a->tokens_per_scene;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (probe == 1) hipLaunchKernelGGL((scene_layer_frag_kernel<true, 1>), dim3(scenes), dim3(512), 0, s, *a);
    else if (probe == 2) hipLaunchKernelGGL((scene_layer_frag_kernel<true, 2>), dim3(scenes), dim3(512), 0, s, *a);
    else if (probe == 3) hipLaunchKernelGGL((scene_layer_frag_kernel<true, 3>), dim3(scenes), dim3(512), 0, s, *a);
    else if (probe == 4) hipLaunchKernelGGL((scene_layer_frag_kernel<true, 4>), dim3(scenes), dim3(512), 0, s, *a);
    else if (gn) hipLaunchKernelGGL((scene_layer_frag_kernel<true, 0>), dim3(scenes), dim3(512), 0, s, *a);
    else hipLaunchKernelGGL((scene_layer_frag_kernel<false, 0>), dim3(scenes), dim3(512), 0, s, *a);
    return (int)hipGetLastError();
}

extern "C" int tune_scene_direct_launch(int gn, const dsc_gemm_args* a, void* stream) {
    const int scenes = a->m / a->tokens_per_scene;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = dsc_scene::SC_WP_SMEM_FLOATS * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scene_layer_direct_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scene_layer_direct_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    if (gn) hipLaunchKernelGGL(scene_layer_direct_kernel<true>, dim3(scenes), dim3(512), lds, s, *a);
    else hipLaunchKernelGGL(scene_layer_direct_kernel<false>, dim3(scenes), dim3(512), lds, s, *a);
    return (int)hipGetLastError();
}

extern "C" int tune_scene_launch(int gn, const dsc_gemm_args* a, void* stream) {
    static bool attr = false;
    const size_t lds = dsc_scene::SC_SMEM_FLOATS * sizeof(float);
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(scene_layer_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(scene_layer_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    const int scenes = a->m / a->tokens_per_scene;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (gn) hipLaunchKernelGGL(scene_layer_kernel<true>, dim3(scenes), dim3(512), lds, s, *a);
    else hipLaunchKernelGGL(scene_layer_kernel<false>, dim3(scenes), dim3(512), lds, s, *a);
    return (int)hipGetLastError();
}

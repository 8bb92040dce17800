// Attribution probes of the split-bf16 weight-gradient kernel (csrc/gemm_tn_split.h): the product's block body with ONE ingredient of
// its token step removed per variant (template parameter PROBE; results are garbage for PROBE != 0).  Harness: tools/tn_probe.py.
#include "../diffuscene_amd/csrc/gemm_tn_split.h"

template <int PROBE>
__global__ __launch_bounds__(512, 1) void tn_probe_kernel(const float* a, const float* dy, float* out, int m, int n, int k, int tiles_per_layer,
                                                          long layer_stride_a, long layer_stride_out, int map) {
    __shared__ __attribute__((aligned(16))) char smem[dsc_tn_split::SMEM];
    // block -> (layer, tile).  map 0: consecutive ids = the tiles of one layer (what the grouped product launch does: the hardware deals
    // consecutive workgroup ids round-robin over the 8 XCDs, so a layer's 8 tiles -- which share its dY / A strips -- land on 8
    // DIFFERENT L2s);  map 1: XCD-grouped -- the ids of one XCD (b % 8) take consecutive (layer, tile) slots, so a layer's tiles share
    // one L2;  map 2: every block works on layer 0 (all strips hot in every L2: the cache-hit bound)
    int lb = blockIdx.x;
    if (map == 1) lb = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    int layer = lb / tiles_per_layer;
    const int local = lb % tiles_per_layer;
    if (map == 2) layer = 0;
    const int ktiles = (k + 127) / 128;
    dsc_tn_split::Prob p;
    p.a1 = a + layer * layer_stride_a; p.lda1 = k; p.k1 = k; p.a2 = dy; p.lda2 = n; p.k2 = 0;   // (a2 unused: k2 == 0; a null constant here crashes clang-22 instcombine)
    p.dy = dy + layer * layer_stride_a; p.ldd = n;
    p.m = m; p.n = n; p.kvalid = k;
    p.chunk = (m + 31) / 32 * 32;
    p.out = out + layer * layer_stride_out; p.ldo = k; p.bias_out = nullptr; p.slab = 0; p.bias_slab = 0;
    dsc_tn_split::tn_split_block<PROBE>(p, local % ktiles, local / ktiles, 0, smem);
}

extern "C" int tn_probe(int probe, const float* a, const float* dy, float* out, int m, int n, int k, int layers, int map, void* stream) {
    const int tiles = ((n + 255) / 256) * ((k + 127) / 128);
    dim3 grid((unsigned)(tiles * layers)), block(512);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long sa = (long)m * k, so = (long)n * k;
    switch (probe) {
        case 0: hipLaunchKernelGGL(tn_probe_kernel<0>, grid, block, 0, s, a, dy, out, m, n, k, tiles, sa, so, map); break;
        case 1: hipLaunchKernelGGL(tn_probe_kernel<1>, grid, block, 0, s, a, dy, out, m, n, k, tiles, sa, so, map); break;
        case 2: hipLaunchKernelGGL(tn_probe_kernel<2>, grid, block, 0, s, a, dy, out, m, n, k, tiles, sa, so, map); break;
        case 3: hipLaunchKernelGGL(tn_probe_kernel<3>, grid, block, 0, s, a, dy, out, m, n, k, tiles, sa, so, map); break;
        case 4: hipLaunchKernelGGL(tn_probe_kernel<4>, grid, block, 0, s, a, dy, out, m, n, k, tiles, sa, so, map); break;
        case 5: hipLaunchKernelGGL(tn_probe_kernel<5>, grid, block, 0, s, a, dy, out, m, n, k, tiles, sa, so, map); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

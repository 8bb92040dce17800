// Round-4 GO / NO-GO experiment: the two Block launches of a ResnetBlock (denoise_net.py:190-206) as ONE persistent
// launch on the product's split-bf16 tile.  Not part of the product library: built into tools/_build/libtwo_layer_probe.so by
// tools/two_layer_probe.py, which also holds the harness (bit-comparison with the two product launches, sustained graph timing).
//
//   block (row tile r, column block c):   layer-1 tile (r, c)  ->  wait until all `cbs` column blocks of row tile r have stored theirs
//                                         ->  layer-2 tile (r, c), whose token rows are layer 1's output rows of row tile r
//
// One block per CU (the tile owns the LDS) and grid == the tile count of ONE layer, so every block is resident and the spin wait cannot
// deadlock as long as the dispatcher hands out workgroups in id order (siblings are 8 ids apart under the XCD block map); a bounded
// spin writes an error flag instead of hanging the GPU.  Layer 2's first weight tile is DMA'd before the wait (gemm_split_tile's sync
// hook).  `fence` selects the memory ordering around the flag: 0 = s_waitcnt vmcnt(0) only (the siblings share one XCD's L2 under the
// block map), 1 = agent-scope release / acquire fences (L2 write-back + L1 invalidate: what a cross-XCD dependency would need).
#include "../diffuscene_amd/csrc/gemm_split.hip"

namespace dsc_split {

struct PairSync {
    unsigned* flag;
    unsigned target;
    unsigned* err;
    int fence;
    __device__ __forceinline__ void operator()() const {
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 22)) {           // ~seconds: give up loudly instead of hanging the box
                    atomicExch(err, 1u);
                    break;
                }
            }
        }
        __syncthreads();
        if (fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
};

template <int WM, int WN, int RB>
__global__ __launch_bounds__(64 * WM * WN, 1) void gemm_split_pair_kernel(const dsc_gemm_args p1, const dsc_gemm_args p2, const int ntok,
                                                                           unsigned* flags, unsigned* err, const int fence) {
    __shared__ __attribute__((aligned(16))) char smem[split_smem_bytes<WM, WN, RB>()];
    gemm_split_tile<true, WM, WN, RB, true>(p1, ntok, blockIdx.x, 0, smem);
    // the row tile of this block, as gemm_split_tile maps it
    constexpr int BN = 64 * WN;
    const int scenes = (p1.m + ntok - 1) / ntok;
    const int cbs = p1.n / BN, rbs = (scenes + WM - 1) / WM;
    int rb;
    if ((rbs & 7) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        rb = xcd * (rbs >> 3) + idx / cbs;
    } else {
        rb = blockIdx.x / cbs;
    }
    // every wave's stores of layer 1 have left (vmcnt counts stores on gfx9), every wave is done with the LDS
    if (fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    unsigned target = 0;
    if (threadIdx.x == 0) {
        // counters only grow: launches are stream-ordered, so a launch starts at a multiple of cbs and ends at the next one
        const unsigned old = __hip_atomic_fetch_add(flags + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        target = (old / (unsigned)cbs + 1u) * (unsigned)cbs;
    }
    gemm_split_tile<true, WM, WN, RB, true>(p2, ntok, blockIdx.x, 0, smem, PairSync{flags + rb, target, err, fence});
}

}  // namespace dsc_split

// the tile the product's dispatcher would take for this launch shape: 65..80 tokens per scene and >= 192 eight-wave blocks -> <2,4,5>
// (headline, B = 256); fewer -> the four-wave <2,2,5> (B = 128: complete / arrange); 17..32 tokens -> <4,2,2> (bedroom21)
template <int WM, int WN, int RB>
static int launch_pair(const dsc_gemm_args* a1, const dsc_gemm_args* a2, unsigned* flags, unsigned* err, int fence, hipStream_t s) {
    const int N = a1->tokens_per_scene;
    const int scenes = (a1->m + N - 1) / N;
    const unsigned grid = (unsigned)(((scenes + WM - 1) / WM) * (a1->n / (64 * WN)));
    if (grid > 256) return DSC_ERANGE;                  // every block must be resident: one block per CU
    hipLaunchKernelGGL((dsc_split::gemm_split_pair_kernel<WM, WN, RB>), dim3(grid), dim3(64 * WM * WN), 0, s, *a1, *a2, N, flags, err, fence);
    return (int)hipGetLastError();
}

extern "C" int probe_gn_pair(const dsc_gemm_args* a1, const dsc_gemm_args* a2, unsigned* flags, unsigned* err, int fence, void* stream) {
    if (!a1 || !a2 || !flags || !err || a1->m != a2->m || a1->n != a2->n || a1->n % 256 || !a1->w_planes || !a2->w_planes) return DSC_EINVAL;
    const int N = a1->tokens_per_scene;
    if (a2->tokens_per_scene != N || a1->batch != 1 || a2->batch != 1) return DSC_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int scenes = (a1->m + N - 1) / N;
    if (N > 64 && N <= 80) {
        if (((scenes + 1) / 2) * (a1->n / 256) >= 192) return launch_pair<2, 4, 5>(a1, a2, flags, err, fence, s);
        return launch_pair<2, 2, 5>(a1, a2, flags, err, fence, s);
    }
    if (N > 16 && N <= 32) return launch_pair<4, 2, 2>(a1, a2, flags, err, fence, s);
    return DSC_EINVAL;
}

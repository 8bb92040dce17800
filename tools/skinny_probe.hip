// Phase stamps and sustained timing of the K-parallel small-launch GEMM (csrc/gemm_skinny.h, round 6).  No torch: a gfx950 binary.
//
//   ./tools/skinny_probe [gn=1] [K=512] [N=21] [scenes=1] [reps=400] [wait=1] [loopform=0]
//
// Launches the kernel directly over R rotating operand sets (weights of R different layers, as a reverse step walks them), prints the sustained
// us per launch from hipEvents and, from s_memtime stamps of every wave of block 0: launch -> operands requested+arrived -> MFMAs done -> partials
// in LDS + barrier -> reduced -> end.  wait=1 forces s_waitcnt vmcnt(0) before the first MFMA (separates load latency from MFMA time);
// wait=0 leaves the product's own waits.
#include <hip/hip_runtime.h>
__device__ unsigned long long* g_skinny_stamps;
__device__ int g_skinny_wait;
#define DSC_SKINNY_STAMP(id) { st_[id] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define DSC_SKINNY_STAMP_LOADED { if (g_skinny_wait) __builtin_amdgcn_s_waitcnt(0x0f70); st_[6] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define DSC_SKINNY_STAMP_END                                                                          \
    st_[5] = __builtin_amdgcn_s_memtime();                                                            \
    if (g_skinny_stamps && blockIdx.x == 0 && (threadIdx.x & 63) == 0) {                              \
        unsigned long long* d_ = g_skinny_stamps + 8 * (threadIdx.x >> 6);                            \
        for (int q_ = 0; q_ < 8; ++q_) d_[q_] = st_[q_];                                              \
    }
#define DSC_SKINNY_STAMP_DECL unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; st_[7] = __builtin_amdgcn_s_memtime();
#include "gemm_skinny.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                         \
    do {                                                                                              \
        hipError_t e_ = (x);                                                                          \
        if (e_ != hipSuccess) {                                                                       \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));         \
            exit(2);                                                                                  \
        }                                                                                             \
    } while (0)

static unsigned rng_state = 12345u;
static float frand() {
    rng_state = rng_state * 1664525u + 1013904223u;
    return ((rng_state >> 8) * (1.0f / 8388608.0f)) - 1.0f;
}
static float* dev_random(size_t n, float scale) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = frand() * scale;
    float* d;
    CK(hipMalloc(&d, n * 4));
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

template <bool GN, int NG>
static void run(const dsc_gemm_args* sets, int R, int reps, int ncb, int rpb, int nrb, hipStream_t s, const char* what) {
    using dsc_skinny::gemm_skinny_kernel;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_skinny_kernel<GN, NG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                           (int)dsc_skinny::lds_bytes<NG>()));
    for (int i = 0; i < 20; ++i)
        hipLaunchKernelGGL((gemm_skinny_kernel<GN, NG>), dim3(nrb * ncb), dim3(512), dsc_skinny::lds_bytes<NG>(), s, sets[i % R], ncb, rpb);
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i)
            hipLaunchKernelGGL((gemm_skinny_kernel<GN, NG>), dim3(nrb * ncb), dim3(512), dsc_skinny::lds_bytes<NG>(), s, sets[i % R], ncb, rpb);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.2f us per launch (%d launches over %d operand sets)\n", what, ms * 1000.0f / reps, reps, R);
    }
}

int main(int argc, char** argv) {
    const int gn = argc > 1 ? atoi(argv[1]) : 1, K = argc > 2 ? atoi(argv[2]) : 512, N = argc > 3 ? atoi(argv[3]) : 21;
    const int scenes = argc > 4 ? atoi(argv[4]) : 1, reps = argc > 5 ? atoi(argv[5]) : 400, wait = argc > 6 ? atoi(argv[6]) : 1;
    const int loopform = argc > 7 ? atoi(argv[7]) : 0;       // 1: also time the two-buffer loop form (its stamps then replace the static form's)
    const int M = scenes * N, n = 512, R = 56;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    std::vector<dsc_gemm_args> sets(R);
    float* a = dev_random((size_t)M * K, 1.0f);
    float* res = dev_random((size_t)M * n, 1.0f);
    float* ss = dev_random((size_t)scenes * 2 * n, 0.3f);
    float *y;
    CK(hipMalloc(&y, (size_t)M * n * 4));
    for (int r = 0; r < R; ++r) {
        dsc_gemm_args g = {};
        g.a1 = a; g.lda1 = K; g.k1 = K;
        g.w = dev_random((size_t)n * K, 0.06f); g.ldw = K;
        g.bias = dev_random(n, 1.0f);
        g.residual = res; g.ldr = n;
        g.y = y; g.ldy = n; g.m = M; g.n = n; g.batch = 1;
        if (gn) {
            g.gamma = dev_random(n, 1.0f); g.beta = dev_random(n, 1.0f); g.eps = 1e-5f; g.tokens_per_scene = N;
            g.scale_shift = ss; g.ld_ss = 2 * n; g.ss_mode = DSC_SS_PER_SCENE;
        }
        sets[r] = g;
    }
    const int rpb = (32 / N) * N > M ? M : (32 / N) * N, nrb = (M + rpb - 1) / rpb, ncb = n / 64;
    unsigned long long* stamps;
    CK(hipMalloc(&stamps, 64 * 8));
    CK(hipMemset(stamps, 0, 64 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_skinny_stamps), &stamps, sizeof(stamps)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_skinny_wait), &wait, sizeof(wait)));
    printf("M=%d (N=%d x %d scenes) n=%d K=%d gn=%d: %d blocks of 512 threads, forced wait %d\n", M, N, scenes, n, K, gn, nrb * ncb, wait);
    const int ng = K <= 256 ? 1 : K <= 512 ? 2 : K <= 1024 ? 4 : 0;
    const int direct = argc > 8 ? atoi(argv[8]) : 0;         // 1: time the direct-fragment forms even where the staged form applies
    if ((K % 512) == 0 && !direct) {
        if (gn) run<true, dsc_skinny::STAGED>(sets.data(), R, reps, ncb, rpb, nrb, s, "GN staged");
        else run<false, dsc_skinny::STAGED>(sets.data(), R, reps, ncb, rpb, nrb, s, "plain staged");
    } else if (gn) {
        if (ng == 1) run<true, 1>(sets.data(), R, reps, ncb, rpb, nrb, s, "GN NG=1");
        if (ng == 2) run<true, 2>(sets.data(), R, reps, ncb, rpb, nrb, s, "GN NG=2");
        if (ng == 4) run<true, 4>(sets.data(), R, reps, ncb, rpb, nrb, s, "GN NG=4");
        if (ng == 0 || loopform) run<true, 0>(sets.data(), R, reps, ncb, rpb, nrb, s, "GN NG=0 (two buffers in flight)");
    } else {
        if (ng == 1) run<false, 1>(sets.data(), R, reps, ncb, rpb, nrb, s, "plain NG=1");
        if (ng == 2) run<false, 2>(sets.data(), R, reps, ncb, rpb, nrb, s, "plain NG=2");
        if (ng == 4) run<false, 4>(sets.data(), R, reps, ncb, rpb, nrb, s, "plain NG=4");
        if (ng == 0 || loopform) run<false, 0>(sets.data(), R, reps, ncb, rpb, nrb, s, "plain NG=0 (two buffers in flight)");
    }
    std::vector<unsigned long long> h(64);
    CK(hipMemcpy(h.data(), stamps, 64 * 8, hipMemcpyDeviceToHost));
    printf("stamps of the last launch, block 0 (s_memtime ticks since kernel entry): loaded | mma done | partials+barrier | reduced | end\n");
    for (int w = 0; w < 8; ++w) {
        const unsigned long long* d = h.data() + 8 * w;
        printf("  wave %d: entry->K section %6lld | loaded %6lld | mma %6lld | barrier %6lld | reduced %6lld | end %6lld\n", w, (long long)(d[0] - d[7]),
               (long long)(d[6] - d[7]), (long long)(d[1] - d[7]), (long long)(d[2] - d[7]), (long long)(d[3] - d[7]), (long long)(d[5] - d[7]));
    }
    return 0;
}

// Harness around an INSTRUMENTED copy of the split-bf16 GEMM (tools/split_probe_gen.py inserts s_memtime stamps into the text of
// csrc/gemm_split.hip at build time).  No torch: a gfx950 binary that links libdiffuscene_hip.so for the product launch it is compared with.
//
//   ./tools/split_probe [gn=1] [res=0] [roles=0] [K=512] [reps=200]      roles = 1: the role-split K loop (template parameter ROLES)
//
// Prints: bit comparison instrumented vs product output, sustained us per launch (product / instrumented with stamps off / on, rotating
// over operand sets larger than the Infinity Cache), the shader clock during the run, and cycles per phase of a K tile:
//   wait   before the top-of-tile s_waitcnt -> after the block barrier     (own DMA / plane writes landed, slowest wave arrived)
//   top    barrier -> first MFMA block (fragment reads issued; the stamp itself waits for them)
//   blk i  the MFMA block of token rows 16 i .. 16 i + 15 (24 MFMAs; the last blocks carry the split of the next tile)
#include "dsc_common.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static unsigned long long* g_stamps = nullptr;
static int g_stamp_stride = 1;

#define NST 8
#define KTMAX 32
#define SLOTS 8
#define PROBE_WAVES 8
#define PROBE_TAIL (SLOTS * PROBE_WAVES * KTMAX * NST)

#ifdef PROBE_NOSTAMPS            /* plain copy of the kernel (A/B of spliced variants without the stamps' waits) */
#define PROBE_DECL
#define PROBE_STAMP(id)
#define PROBE_STAMP_BLOCK(i)
#define PROBE_FLUSH
#define PROBE_END
#else
#define PROBE_DECL                                                                                                             \
    unsigned long long st_[NST] = {0, 0, 0, 0, 0, 0, 0, 0};                                                                   \
    const bool st_on_ = stamps && (blockIdx.x % stamp_stride) == 0 && (int)(blockIdx.x / stamp_stride) < SLOTS;               \
    const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime(), ct0_ = __builtin_amdgcn_s_memtime();
#define PROBE_STAMP(id)                                                                                                        \
    st_[id] = __builtin_amdgcn_s_memtime();                                                                                    \
    __builtin_amdgcn_sched_barrier(0);
#define PROBE_STAMP_BLOCK(i)                                                                                                   \
    st_[3 + (i)] = __builtin_amdgcn_s_memtime();                                                                               \
    __builtin_amdgcn_sched_barrier(0);
#define PROBE_FLUSH                                                                                                            \
    if (st_on_ && lane == 0) {                                                                                                 \
        unsigned long long* d_ = stamps + ((((size_t)(blockIdx.x / stamp_stride) * PROBE_WAVES + wave_u) * KTMAX + kt) * NST); \
        for (int q_ = 0; q_ < NST; ++q_) d_[q_] = st_[q_];                                                                     \
    }
#define PROBE_END                                                                                                              \
    if (st_on_ && lane == 0 && wave_u == 0) {                                                                                  \
        unsigned long long* d_ = stamps + PROBE_TAIL + 2 * (blockIdx.x / stamp_stride);                                       \
        d_[0] = __builtin_amdgcn_s_memrealtime() - rt0_;                                                                       \
        d_[1] = __builtin_amdgcn_s_memtime() - ct0_;                                                                           \
    }

#endif

#ifndef DSC_SPLIT_DSPREAD
#define DSC_SPLIT_DSPREAD true
#endif
#include "split_probe_kernel.inc"

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

int main(int argc, char** argv) {
    const int gn = argc > 1 ? atoi(argv[1]) : 1, res = argc > 2 ? atoi(argv[2]) : 0, roles = argc > 3 ? atoi(argv[3]) : 0;
    const int K = argc > 4 ? atoi(argv[4]) : 512, reps = argc > 5 ? atoi(argv[5]) : 200;
    const int M = 20480, n = 512, ntok = 80, scenes = M / ntok, R = 6;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    float *A[R], *Y[R], *RS[R];
    for (int r = 0; r < R; ++r) {
        A[r] = dev_random((size_t)M * K, 1.0f);
        RS[r] = dev_random((size_t)M * n, 1.0f);
        CK(hipMalloc(&Y[r], (size_t)M * n * 4));
    }
    float* W = dev_random((size_t)n * K, 0.05f);
    float* bias = dev_random(n, 0.1f);
    float* gamma = dev_random(n, 1.0f);
    float* beta = dev_random(n, 0.5f);
    float* ss = dev_random((size_t)scenes * 2 * n, 0.3f);
    uint16_t* planes;
    CK(hipMalloc(&planes, (size_t)3 * n * K * 2));
    dsc_split_item it = {W, K, n, K, planes, 0};
    if (dsc_split_bf16x3_f32(&it, 1, s) != 0) { fprintf(stderr, "split failed\n"); return 2; }
    unsigned long long* stamps;
    const size_t stamp_words = PROBE_TAIL + 2 * SLOTS;
    CK(hipMalloc(&stamps, stamp_words * 8));
    CK(hipMemset(stamps, 0, stamp_words * 8));
    float* Yp;
    CK(hipMalloc(&Yp, (size_t)M * n * 4));

    auto args_for = [&](int r, float* y) {
        dsc_gemm_args a;
        memset(&a, 0, sizeof(a));
        a.a1 = A[r]; a.lda1 = K; a.k1 = K;
        a.w = W; a.ldw = K; a.bias = bias;
        a.residual = res ? RS[r] : nullptr; a.ldr = n;
        a.y = y; a.ldy = n; a.m = M; a.n = n; a.batch = 1;
        a.gamma = gamma; a.beta = beta; a.eps = 1e-5f; a.tokens_per_scene = ntok;
        a.scale_shift = ss; a.ld_ss = 2 * n; a.ss_mode = DSC_SS_PER_SCENE;
        a.w_planes = planes;
        return a;
    };
    auto product = [&](const dsc_gemm_args& a) { return gn ? dsc_gemm_gn_silu_f32(&a, s) : dsc_gemm_f32(&a, s); };
    auto probe = [&](const dsc_gemm_args& a) {
        if (roles) return gn ? dsc_split_probe::launch<true, 2, 4, 5, true>(&a, ntok, s) : dsc_split_probe::launch<false, 2, 4, 5, true>(&a, ntok, s);
        return gn ? dsc_split_probe::launch<true, 2, 4, 5>(&a, ntok, s) : dsc_split_probe::launch<false, 2, 4, 5>(&a, ntok, s);
    };
    {   // the product launch must be on the split kernel, and the instrumented copy must compute the same bits
        dsc_gemm_args a = args_for(0, Y[0]);
        printf("dsc_gemm_arithmetic = %d (1 = split-bf16)\n", dsc_gemm_arithmetic(&a, gn));
        if (product(a) != 0) { fprintf(stderr, "product launch failed\n"); return 2; }
        dsc_gemm_args b = args_for(0, Yp);
        g_stamps = stamps; g_stamp_stride = 32;
        if (probe(b) != 0) { fprintf(stderr, "probe launch failed\n"); return 2; }
        CK(hipStreamSynchronize(s));
        std::vector<float> h0((size_t)M * n), h1((size_t)M * n);
        CK(hipMemcpy(h0.data(), Y[0], h0.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h1.data(), Yp, h1.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < h0.size(); ++i) bad += memcmp(&h0[i], &h1[i], 4) != 0;
        printf("instrumented vs product output: %zu of %zu elements differ (y[0] = %g)\n", bad, h0.size(), h0[0]);
    }
    auto sustained = [&](int which) {          // 0 product, 1 instrumented (stamps off), 2 instrumented (stamps on)
        g_stamps = which == 2 ? stamps : nullptr;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int i = 0; i < 100; ++i) { dsc_gemm_args a = args_for(i % R, Y[i % R]); which ? probe(a) : product(a); }   // clocks settle
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) { dsc_gemm_args a = args_for(i % R, Y[i % R]); which ? probe(a) : product(a); }
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1000.f / reps;
    };
    const float t0 = sustained(0), t1 = sustained(1), t2 = sustained(2), t0b = sustained(0);
    printf("gn=%d res=%d roles=%d K=%d: product %.2f us (again %.2f), instrumented %.2f us (stamps off) %.2f us (stamps on)\n", gn, res, roles, K, t0, t0b, t1, t2);

#ifdef PROBE_NOSTAMPS
    return 0;
#endif
    // the stamps of the LAST launch of the sustained run
    std::vector<unsigned long long> h(stamp_words);
    CK(hipMemcpy(h.data(), stamps, stamp_words * 8, hipMemcpyDeviceToHost));
    const int KT = K / 32, RB = 5;
    double ghz = 0;
    for (int sl = 0; sl < SLOTS; ++sl) ghz += (double)h[PROBE_TAIL + 2 * sl + 1] / (double)h[PROBE_TAIL + 2 * sl] * 0.1 / SLOTS;
    printf("shader clock %.3f GHz (s_memtime / s_memrealtime at 100 MHz); kernel %.0f cycles\n", ghz, (double)h[PROBE_TAIL + 1]);
    const char* names[3 + 5] = {"wait", "top", "blk0", "blk1", "blk2", "blk3", "blk4", "flush"};
    double tot[8] = {0}, mx[8] = {0};
    long cnt = 0;
    double tile_sum = 0;
    printf("%-12s %7s %7s %7s %7s %7s %7s %7s %7s %8s\n", "slot.wave", names[0], names[1], names[2], names[3], names[4], names[5], names[6], names[7], "tile");
    for (int sl = 0; sl < SLOTS; ++sl)
        for (int w = 0; w < PROBE_WAVES; ++w) {
            double seg[8] = {0};
            int c = 0;
            for (int kt = 2; kt < KT - 2; ++kt) {
                const unsigned long long* d = &h[(((size_t)sl * PROBE_WAVES + w) * KTMAX + kt) * NST];
                const unsigned long long* nx = d + NST;
                if (!d[0] || !nx[0]) continue;
                seg[0] += (double)(d[1] - d[0]);
                seg[1] += (double)(d[2] - d[1]);
                for (int i = 0; i < RB; ++i) seg[2 + i] += (double)(d[3 + i] - d[2 + i]);
                seg[7] += (double)(nx[0] - d[2 + RB]);
                ++c;
            }
            if (!c) continue;
            double tile = 0;
            for (int q = 0; q < 8; ++q) { seg[q] /= c; tile += seg[q]; tot[q] += seg[q]; mx[q] = std::max(mx[q], seg[q]); }
            tile_sum += tile;
            ++cnt;
            if (sl < 2) printf("%d.%d          %7.0f %7.0f %7.0f %7.0f %7.0f %7.0f %7.0f %7.0f %8.0f\n", sl, w, seg[0], seg[1], seg[2], seg[3], seg[4], seg[5], seg[6], seg[7], tile);
        }
    if (cnt) {
        printf("%-12s", "mean");
        for (int q = 0; q < 8; ++q) printf(" %7.0f", tot[q] / cnt);
        printf(" %8.0f\n%-12s", tile_sum / cnt, "max");
        for (int q = 0; q < 8; ++q) printf(" %7.0f", mx[q]);
        printf("\n(%ld waves; MFMA issue floor per tile = 2 waves x %d MFMAs x 16 cycles = %d cycles per SIMD)\n", cnt, RB * 24, 2 * RB * 24 * 16);
    }
    return 0;
}

// GO / NO-GO harness of the wave-autonomous split-bf16 GEMM (csrc/gemm_split_wave.h, round 6) against the product's block-staged kernel
// (csrc/gemm_split.hip behind dsc_gemm_gn_silu_f32 / dsc_gemm_f32).  No torch: a gfx950 binary that links libdiffuscene_hip.so.
//
//   ./tools/wave_probe [gn=1] [res=0] [K=512] [reps=300] [scenes=256]
//
// Prints the bit comparison with the product launch and the sustained us per launch of both, alternating, over operand sets larger than
// the Infinity Cache.
#include <hip/hip_runtime.h>
// stamps: per sampled block (every 32nd) and wave, per K tile: top of tile, after each token block; per kernel: start, loop start, loop end, end
static unsigned long long* g_stamps = nullptr;
#define ST_N 8
#define ST_KT 32
#define ST_SLOTS 8
#define ST_TAIL (ST_SLOTS * 4 * ST_KT * ST_N)
#define DSC_WAVE_STAMP_ARGS , unsigned long long* const stamps
#define DSC_WAVE_STAMP_PASS , g_stamps
#define DSC_WAVE_STAMP_DECL                                                                                              \
    unsigned long long st_[ST_N] = {0, 0, 0, 0, 0, 0, 0, 0}, sk_[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                      \
    const bool st_on_ = stamps && (blockIdx.x % 32) == 0 && (int)(blockIdx.x / 32) < ST_SLOTS;                           \
    const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime();
#define DSC_WAVE_STAMP_KERNEL(id) sk_[id] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
#define DSC_WAVE_STAMP_TILE(id) st_[id] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
#define DSC_WAVE_STAMP_FLUSH(kt)                                                                                         \
    if (st_on_ && lane == 0) {                                                                                           \
        unsigned long long* d_ = stamps + ((((size_t)(blockIdx.x / 32) * 4 + wave) * ST_KT + (kt)) * ST_N);              \
        for (int q_ = 0; q_ < ST_N; ++q_) d_[q_] = st_[q_];                                                              \
    }
#define DSC_WAVE_STAMP_END                                                                                               \
    sk_[7] = __builtin_amdgcn_s_memtime();                                                                               \
    if (st_on_ && lane == 0) {                                                                                           \
        unsigned long long* d_ = stamps + ST_TAIL + 16 * ((blockIdx.x / 32) * 4 + wave);                                  \
        for (int q_ = 0; q_ < 8; ++q_) d_[q_] = sk_[q_];                                                                \
        d_[8] = __builtin_amdgcn_s_memrealtime() - rt0_;                                                                 \
    }
#include "gemm_split_wave.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

int main(int argc, char** argv) {
    const int gn = argc > 1 ? atoi(argv[1]) : 1, res = argc > 2 ? atoi(argv[2]) : 0;
    const int K = argc > 3 ? atoi(argv[3]) : 512, reps = argc > 4 ? atoi(argv[4]) : 300;
    const int scenes = argc > 5 ? atoi(argv[5]) : 256, ntok = argc > 6 ? atoi(argv[6]) : 80;
    const int ssmode = argc > 7 ? atoi(argv[7]) : DSC_SS_PER_SCENE, pre = argc > 8 ? atoi(argv[8]) : 0, jb = argc > 9 ? atoi(argv[9]) : 8;
    const int M = scenes * ntok, n = 512, R = 6;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    dsc_set_split_wave(0);             // "product" below = the block-staged kernel through the library (row-major planes); the wave kernel is launched directly
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
    float* ss = dev_random((size_t)M * 2 * n, 0.3f);
    float *PA, *PB;
    CK(hipMalloc(&PA, (size_t)M * n * 4));
    CK(hipMalloc(&PB, (size_t)M * n * 4));
    uint16_t *planes, *planes_f;
    const size_t pe = (size_t)3 * n * K;
    CK(hipMalloc(&planes, pe * 2));
    CK(hipMalloc(&planes_f, pe * 2));
    dsc_split_item it = {W, K, n, K, planes, 0};
    if (dsc_split_bf16x3_f32(&it, 1, s) != 0) { fprintf(stderr, "split failed\n"); return 2; }
    CK(hipStreamSynchronize(s));
    {   // fragment-major re-layout on the host: [3][n/16][K/32][lane = g*16 + l15][8]
        std::vector<uint16_t> h(pe), f(pe);
        CK(hipMemcpy(h.data(), planes, pe * 2, hipMemcpyDeviceToHost));
        const int KT = K / 32;
        for (int p = 0; p < 3; ++p)
            for (int nb = 0; nb < n / 16; ++nb)
                for (int kt = 0; kt < KT; ++kt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e)
                            f[((((size_t)p * (n / 16) + nb) * KT + kt) * 64 + lane) * 8 + e] =
                                h[(size_t)p * n * K + (size_t)(nb * 16 + (lane & 15)) * K + kt * 32 + (lane >> 4) * 8 + e];
        CK(hipMemcpy(planes_f, f.data(), pe * 2, hipMemcpyHostToDevice));
    }
    float* Yp;
    CK(hipMalloc(&Yp, (size_t)M * n * 4));
    CK(hipMemset(Yp, 0xff, (size_t)M * n * 4));

    auto args_for = [&](int r, float* y) {
        dsc_gemm_args a;
        memset(&a, 0, sizeof(a));
        a.a1 = A[r]; a.lda1 = K; a.k1 = K;
        a.w = W; a.ldw = K; a.bias = bias;
        a.residual = res ? RS[r] : nullptr; a.ldr = n;
        a.y = y; a.ldy = n; a.m = M; a.n = n; a.batch = 1;
        a.gamma = gamma; a.beta = beta; a.eps = 1e-5f; a.tokens_per_scene = ntok;
        a.scale_shift = ss; a.ld_ss = 2 * n; a.ss_mode = ssmode;
        if (pre) a.preact = (y == Yp) ? PB : PA, a.ld_preact = n;
        a.w_planes = planes;
        return a;
    };
    auto product = [&](const dsc_gemm_args& a) { return gn ? dsc_gemm_gn_silu_f32(&a, s) : dsc_gemm_f32(&a, s); };
    auto probe = [&](const dsc_gemm_args& a) {
        const int rb = (ntok + 15) / 16;
        const bool pr = ssmode == DSC_SS_PER_TOKEN || ssmode == DSC_SS_PER_SLOT;
        if (gn && jb == 4 && rb == 5) return pr ? dsc_wave::launch<true, 5, true, 4>(&a, planes_f, ntok, s) : dsc_wave::launch<true, 5, false, 4>(&a, planes_f, ntok, s);
        if (!gn && jb == 4) return dsc_wave::launch<false, 5, false, 4>(&a, planes_f, 80, s);
        if (gn && pr) return rb == 5 ? dsc_wave::launch<true, 5, true>(&a, planes_f, ntok, s) : dsc_wave::launch<true, 2, true>(&a, planes_f, ntok, s);
        if (gn) return rb == 5 ? dsc_wave::launch<true, 5>(&a, planes_f, ntok, s) : rb == 4 ? dsc_wave::launch<true, 4>(&a, planes_f, ntok, s)
                     : rb == 3 ? dsc_wave::launch<true, 3>(&a, planes_f, ntok, s) : dsc_wave::launch<true, 2>(&a, planes_f, ntok, s);
        return dsc_wave::launch<false, 5>(&a, planes_f, 80, s);
    };
    {
        dsc_gemm_args a = args_for(0, Y[0]);
        printf("M=%d n=%d K=%d gn=%d res=%d ntok=%d ss=%d pre=%d: dsc_gemm_arithmetic = %d, tile %d\n", M, n, K, gn, res, ntok, ssmode, pre, dsc_gemm_arithmetic(&a, gn), dsc_gemm_split_tile(&a, gn));
        if (product(a) != 0) { fprintf(stderr, "product launch failed\n"); return 2; }
        dsc_gemm_args b = args_for(0, Yp);
        const int rc = probe(b);
        if (rc != 0) { fprintf(stderr, "probe launch failed: %d\n", rc); return 2; }
        CK(hipStreamSynchronize(s));
        std::vector<float> h0((size_t)M * n), h1((size_t)M * n);
        CK(hipMemcpy(h0.data(), Y[0], h0.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h1.data(), Yp, h1.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0, first = (size_t)-1;
        double maxd = 0;
        for (size_t i = 0; i < h0.size(); ++i)
            if (memcmp(&h0[i], &h1[i], 4) != 0) {
                if (!bad) first = i;
                ++bad;
                maxd = std::max(maxd, (double)fabsf(h0[i] - h1[i]));
            }
        if (pre) {
            std::vector<float> q0((size_t)M * n), q1((size_t)M * n);
            CK(hipMemcpy(q0.data(), PA, q0.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(q1.data(), PB, q1.size() * 4, hipMemcpyDeviceToHost));
            size_t pb = 0;
            for (size_t i = 0; i < q0.size(); ++i) pb += memcmp(&q0[i], &q1[i], 4) != 0;
            printf("saved pre-activation: %zu of %zu elements differ\n", pb, q0.size());
        }
        printf("wave kernel vs product output: %zu of %zu elements differ (max |d| %.3g, first at row %zu col %zu; y[0] = %g / %g)\n", bad, h0.size(), maxd,
               bad ? first / n : 0, bad ? first % n : 0, h0[0], h1[0]);
    }
    auto sustained = [&](int which) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int i = 0; i < 150; ++i) { dsc_gemm_args a = args_for(i % R, Y[i % R]); which ? probe(a) : product(a); }   // clocks settle
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) { dsc_gemm_args a = args_for(i % R, Y[i % R]); which ? probe(a) : product(a); }
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1000.f / reps;
    };
    unsigned long long* stamps;
    const size_t stamp_words = ST_TAIL + 16 * ST_SLOTS * 4;
    CK(hipMalloc(&stamps, stamp_words * 8));
    CK(hipMemset(stamps, 0, stamp_words * 8));
    float t[2][3];
    for (int r = 0; r < 3; ++r)
        for (int w = 0; w < 2; ++w) t[w][r] = sustained(w);
    printf("sustained us per launch: product %.2f %.2f %.2f | wave %.2f %.2f %.2f\n", t[0][0], t[0][1], t[0][2], t[1][0], t[1][1], t[1][2]);
    // stamped run: the last launch of a sustained sequence
    g_stamps = stamps;
    const float ts = sustained(1);
    g_stamps = nullptr;
    std::vector<unsigned long long> h(stamp_words);
    CK(hipMemcpy(h.data(), stamps, stamp_words * 8, hipMemcpyDeviceToHost));
    const int KT = K / 32, RBn = (ntok + 15) / 16;
    printf("with stamps %.2f us.  cycles per K tile (mean over tiles 2..KT-3), per sampled wave: top->blk0 blk1 ... | tile\n", ts);
    double tot[8] = {0}, tile_tot = 0;
    int cnt = 0;
    for (int sl = 0; sl < ST_SLOTS; ++sl)
        for (int w = 0; w < 4; ++w) {
            double seg[8] = {0};
            int c = 0;
            for (int kt = 2; kt < KT - 3; ++kt) {
                const unsigned long long* d = &h[(((size_t)sl * 4 + w) * ST_KT + kt) * ST_N];
                const unsigned long long* nx = d + ST_N;
                if (!d[0] || !nx[0]) continue;
                for (int i = 0; i < RBn; ++i) seg[i] += (double)(d[1 + i] - d[i]);
                seg[RBn] += (double)(nx[0] - d[RBn]);
                ++c;
            }
            if (!c) continue;
            double tile = 0;
            for (int q = 0; q <= RBn; ++q) { seg[q] /= c; tile += seg[q]; tot[q] += seg[q]; }
            tile_tot += tile;
            ++cnt;
            const unsigned long long* k = &h[ST_TAIL + 16 * (sl * 4 + w)];
            if (sl < 3) {
                printf("  %d.%d:", sl, w);
                for (int q = 0; q <= RBn; ++q) printf(" %6.0f", seg[q]);
                printf(" | %6.0f   kernel: prologue %llu  loop %llu  epilogue %llu (stats0 %llu, silu+lds0 %llu, stores0 %llu, cell1 %llu)  total %llu cycles, %.3f GHz\n", tile,
                       k[1] - k[0], k[2] - k[1], k[7] - k[2], k[3] ? k[3] - k[2] : 0, k[3] ? k[4] - k[3] : 0, k[3] ? k[5] - k[4] : 0, k[3] ? k[7] - k[5] : 0, k[7] - k[0],
                       k[8] ? (double)(k[7] - k[0]) / (double)k[8] * 0.1 : 0.0);
            }
        }
    if (cnt) {
        printf("  mean:");
        for (int q = 0; q <= RBn; ++q) printf(" %6.0f", tot[q] / cnt);
        printf(" | %6.0f   (MFMA issue floor per tile: %d x 16 = %d cycles)\n", tile_tot / cnt, RBn * 48, RBn * 48 * 16);
    }
    return 0;
}

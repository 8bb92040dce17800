// Micro-probe (not product): sustained fp32 MFMA rate of the two instruction shapes with the accumulator patterns of the
// product GEMM (32x32x2, 5 accumulators) and the scene kernel (16x16x4, 20 accumulators), 8 waves per CU, no memory traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k16(float* out, int iters) {
    f32x4 acc[5][4];
    for (int a = 0; a < 5; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float wf[4], xf[5];
    for (int i = 0; i < 4; ++i) wf[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 5; ++i) xf[i] = threadIdx.x * 0.002f - i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int tt = 0; tt < 5; ++tt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    acc[tt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ct], xf[tt], acc[tt][ct], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < 5; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k32(float* out, int iters) {
    f32x16 acc[5];
    for (int a = 0; a < 5; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float wf = threadIdx.x * 0.001f, xf[5];
    for (int i = 0; i < 5; ++i) xf[i] = threadIdx.x * 0.002f - i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int tm = 0; tm < 5; ++tm) acc[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf, xf[tm], acc[tm], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < 5; ++a) s += acc[a][0] + acc[a][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* out;
    (void)hipMalloc(&out, 1024 * 512 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch, double flop_per_iter_per_wave, int waves) {
        const int iters = 4000;
        launch(100);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        launch(iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double tf = flop_per_iter_per_wave * iters * waves * 256.0 / (ms * 1e-3) / 1e12;
        printf("%-52s %8.3f ms  %7.1f TFLOP/s\n", name, ms, tf);
    };
    run("16x16x4, 20 acc, 8 waves/CU (2 per SIMD)", [&](int it) { hipLaunchKernelGGL(k16<8>, dim3(256), dim3(512), 0, 0, out, it); }, 80.0 * 2048, 8);
    run("16x16x4, 20 acc, 4 waves/CU (1 per SIMD)", [&](int it) { hipLaunchKernelGGL(k16<4>, dim3(256), dim3(256), 0, 0, out, it); }, 80.0 * 2048, 4);
    run("32x32x2,  5 acc, 8 waves/CU (2 per SIMD)", [&](int it) { hipLaunchKernelGGL(k32<8>, dim3(256), dim3(512), 0, 0, out, it); }, 20.0 * 4096, 8);
    run("32x32x2,  5 acc, 4 waves/CU (1 per SIMD)", [&](int it) { hipLaunchKernelGGL(k32<4>, dim3(256), dim3(256), 0, 0, out, it); }, 20.0 * 4096, 4);
    return 0;
}

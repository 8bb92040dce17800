// Micro-probe (not product): SUSTAINED bf16 MFMA rate on random operands (the chip clocks to its power budget: zero-filled or smooth
// operands run faster than random mantissas), v_mfma_f32_16x16x32_bf16 vs v_mfma_f32_32x32x16_bf16, with the accumulator pattern of
// the split GEMM (20 / 5 accumulators per wave, 12 + 15 operand fragments), 8 waves per CU, no memory traffic in the loop.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_probe.hip -o /tmp/mfma_bf16_probe && /tmp/mfma_bf16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 1) void k16(const u32x4* __restrict__ src, float* out, int iters) {
    f32x4 acc[5][4];
    for (int a = 0; a < 5; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 wf[4][3], xf[5][3];
    for (int j = 0; j < 4; ++j) for (int p = 0; p < 3; ++p) wf[j][p] = __builtin_bit_cast(bf16x8, src[(threadIdx.x * 27 + j * 3 + p) & 16383]);
    for (int i = 0; i < 5; ++i) for (int p = 0; p < 3; ++p) xf[i][p] = __builtin_bit_cast(bf16x8, src[(threadIdx.x * 27 + 12 + i * 3 + p) & 16383]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][2], xf[i][0], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], xf[i][2], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][1], xf[i][1], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][1], xf[i][0], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], xf[i][1], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], xf[i][0], acc[i][j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 5; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(512, 1) void k32(const u32x4* __restrict__ src, float* out, int iters) {
    f32x16 acc[5];
    for (int a = 0; a < 5; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 wf[2][3], xf[5][3];
    for (int j = 0; j < 2; ++j) for (int p = 0; p < 3; ++p) wf[j][p] = __builtin_bit_cast(bf16x8, src[(threadIdx.x * 21 + j * 3 + p) & 16383]);
    for (int i = 0; i < 5; ++i) for (int p = 0; p < 3; ++p) xf[i][p] = __builtin_bit_cast(bf16x8, src[(threadIdx.x * 21 + 6 + i * 3 + p) & 16383]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) {
                const int pw = (pr == 0) ? 2 : (pr == 1 || pr == 4 || pr == 5) ? 0 : 1;
                const int px = (pr == 0 || pr == 3 || pr == 5) ? 0 : (pr == 1) ? 2 : 1;
#pragma unroll
                for (int i = 0; i < 5; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk][pw], xf[i][px], acc[i], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int a = 0; a < 5; ++a) s += acc[a][0] + acc[a][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* out;
    u32x4* src;
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMalloc(&src, 16384 * 16);
    for (int mode = 0; mode < 2; ++mode) {
        unsigned* h = (unsigned*)malloc(16384 * 16);
        for (int i = 0; i < 16384 * 4; ++i) {
            // random bf16 pairs with exponents around 1.0 (sign + 8-bit exponent near 127 + 7 random mantissa bits), or zeros
            const unsigned a = 0x3f00u | (rand() & 0x80ff), b = 0x3f00u | (rand() & 0x80ff);
            h[i] = mode == 0 ? (a | (b << 16)) : 0u;
        }
        (void)hipMemcpy(src, h, 16384 * 16, hipMemcpyHostToDevice);
        free(h);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        auto run = [&](const char* name, auto launch, double flop_per_iter_per_wave) {
            const int iters = 6000;                     // ~0.2-0.4 s: the power management has settled
            launch(300);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            launch(iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%-30s %-44s %9.2f ms  %7.1f TFLOP/s\n", mode == 0 ? "random mantissas" : "zero operands", name, ms,
                   flop_per_iter_per_wave * iters * 8 * 256.0 / (ms * 1e-3) / 1e12);
        };
        run("16x16x32 bf16, 20 acc, 120 MFMA per iteration", [&](int it) { hipLaunchKernelGGL(k16, dim3(256), dim3(512), 0, 0, src, out, it); }, 120.0 * 16384);
        run("32x32x16 bf16,  5 acc,  60 MFMA per iteration", [&](int it) { hipLaunchKernelGGL(k32, dim3(256), dim3(512), 0, 0, src, out, it); }, 60.0 * 32768);
    }
    return 0;
}

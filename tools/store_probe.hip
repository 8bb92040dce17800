// Micro-probe (not product): how fast can 512 blocks write a [20480][512] fp32 matrix in the GEMM epilogue's access
// pattern (each wave: 8 rows x 128 contiguous bytes per instruction, 160x128 tile per block), vs a linear stream, with
// plain / nontemporal stores, and the same with a residual read.  Build: hipcc --offload-arch=gfx950 -O3 -o store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 plain, 1 nontemporal, 2 plain + residual read
__global__ __launch_bounds__(256, 2) void tile_store(float* __restrict__ y, const float* __restrict__ r, int ld, int ncb) {
    const int blk = blockIdx.x;
    const int rb = blk / ncb, cb = blk % ncb;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tr = lane >> 3, cq = lane & 7;
    const long row0 = (long)rb * 160;
    const int c = cb * 128 + wave * 32 + cq * 4;
#pragma unroll
    for (int tm = 0; tm < 5; ++tm)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long tok = row0 + tm * 32 + tr + 8 * i;
            f32x4 v = {1.f * tm, 2.f, 3.f, (float)i};
            if (MODE == 2) { const f32x4 q = *reinterpret_cast<const f32x4*>(r + tok * ld + c); v += q; }
            if (MODE == 1) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(y + tok * ld + c));
            else *reinterpret_cast<f32x4*>(y + tok * ld + c) = v;
        }
}

__global__ __launch_bounds__(256) void linear_store(float* __restrict__ y, long n4) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (; i < n4; i += stride) reinterpret_cast<f32x4*>(y)[i] = f32x4{1.f, 2.f, 3.f, 4.f};
}

int main() {
    const int M = 20480, N = 512;
    float *y, *r;
    hipMalloc(&y, (size_t)M * N * 4 * 8);      // 8 rotating buffers so consecutive launches do not hit the same L2 lines
    hipMalloc(&r, (size_t)M * N * 4 * 8);
    hipMemset(r, 0, (size_t)M * N * 4 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto fn) {
        for (int i = 0; i < 8; ++i) fn(i % 8);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int reps = 40;
        for (int i = 0; i < reps; ++i) fn(i % 8);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps;
        printf("%-44s %7.2f us   %.2f TB/s (42 MB written)\n", name, us, (double)M * N * 4 / us / 1e6);
    };
    const size_t off = (size_t)M * N;
    timeit("tile pattern, plain stores", [&](int b) { hipLaunchKernelGGL(tile_store<0>, dim3(512), dim3(256), 0, 0, y + b * off, r, N, 4); });
    timeit("tile pattern, nontemporal stores", [&](int b) { hipLaunchKernelGGL(tile_store<1>, dim3(512), dim3(256), 0, 0, y + b * off, r, N, 4); });
    timeit("tile pattern, residual read + store", [&](int b) { hipLaunchKernelGGL(tile_store<2>, dim3(512), dim3(256), 0, 0, y + b * off, r + ((b + 3) % 8) * off, N, 4); });
    timeit("linear stream, 2048 blocks", [&](int b) { hipLaunchKernelGGL(linear_store, dim3(2048), dim3(256), 0, 0, y + b * off, (long)M * N / 4); });
    timeit("tile pattern, same buffer every launch", [&](int) { hipLaunchKernelGGL(tile_store<0>, dim3(512), dim3(256), 0, 0, y, r, N, 4); });
    return 0;
}

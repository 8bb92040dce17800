// Optimizer step of train_on_batch: global-norm gradient clipping + Adam over all 77.7 M parameters.
// The reference runs torch.nn.utils.clip_grad_norm_ (max 10, diffusion_scene_layout_ddpm.py:465) and torch.optim.Adam
// (networks/__init__.py:29-30): ~30 multi-tensor launches, 2-3 passes over the gradients.  Here: one sum-of-squares
// sweep, a one-block finaliser that leaves the norm and the clip coefficient ON THE DEVICE (no host sync), and one
// Adam sweep that applies the coefficient while it reads the gradient -- 28 B/parameter of HBM traffic + 4 B for the norm.
//
// Work list = device array of chunks (<= 32768 contiguous elements of one parameter each); block per chunk.
// Arithmetic = torch.optim.Adam (foreach path): m += (g - m)(1 - b1); v = v b2 + (1 - b2) g g;
// p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps); L2 weight decay folded into g as torch does.
#include "dsc_common.h"

namespace {

__device__ __forceinline__ bool chunk_vec_ok(const dsc_optim_chunk& c) {
    return (((uintptr_t)c.param | (uintptr_t)c.grad | (uintptr_t)c.exp_avg | (uintptr_t)c.exp_avg_sq) & 15) == 0 &&
           (c.count & 3) == 0;
}

__global__ __launch_bounds__(256) void grad_sumsq_kernel(const dsc_optim_chunk* __restrict__ chunks,
                                                        double* __restrict__ partial) {
    __shared__ double red[4];
    const dsc_optim_chunk c = chunks[blockIdx.x];
    float s0 = 0.f, s1 = 0.f;
    if ((((uintptr_t)c.grad) & 15) == 0 && (c.count & 3) == 0) {
        const f32x4* g4 = reinterpret_cast<const f32x4*>(c.grad);
        const long n4 = c.count >> 2;
        for (long i = threadIdx.x; i < n4; i += 256) {
            const f32x4 g = g4[i];
            s0 += g[0] * g[0] + g[1] * g[1];
            s1 += g[2] * g[2] + g[3] * g[3];
        }
    } else {
        for (long i = threadIdx.x; i < c.count; i += 256) { const float g = c.grad[i]; s0 += g * g; }
    }
    double s = (double)s0 + (double)s1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// total_norm = sqrt(sum partial); clip_coef = min(1, max_norm / (total_norm + 1e-6))   (torch clip_grad_norm_)
__global__ __launch_bounds__(256) void clip_coef_kernel(const double* __restrict__ partial, int n, float max_norm,
                                                       float* __restrict__ total_norm, float* __restrict__ clip_coef) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float nrm = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
        *total_norm = nrm;
        const float coef = max_norm / (nrm + 1e-6f);
        *clip_coef = coef < 1.0f ? coef : 1.0f;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(const dsc_optim_chunk* __restrict__ chunks, float step_size, float beta1,
                                                  float beta2, float bc2_sqrt, float eps, float weight_decay,
                                                  const float* __restrict__ grad_scale) {
    const dsc_optim_chunk c = chunks[blockIdx.x];
    const float gs = grad_scale ? *grad_scale : 1.0f;
    const float omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
    auto upd = [&](float& p, float g, float& m, float& v) {
        g = g * gs;
        if (weight_decay != 0.f) g = g + weight_decay * p;
        m = m + (g - m) * omb1;
        v = v * beta2 + omb2 * g * g;
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        p = p - step_size * (m / denom);
    };
    if (chunk_vec_ok(c)) {
        f32x4* p4 = reinterpret_cast<f32x4*>(c.param);
        const f32x4* g4 = reinterpret_cast<const f32x4*>(c.grad);
        f32x4* m4 = reinterpret_cast<f32x4*>(c.exp_avg);
        f32x4* v4 = reinterpret_cast<f32x4*>(c.exp_avg_sq);
        const long n4 = c.count >> 2;
        for (long i = threadIdx.x; i < n4; i += 256) {
            f32x4 p = p4[i], m = m4[i], v = v4[i];
            const f32x4 g = g4[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = p[e], me = m[e], ve = v[e];
                upd(pe, g[e], me, ve);
                p[e] = pe; m[e] = me; v[e] = ve;
            }
            p4[i] = p; m4[i] = m; v4[i] = v;
        }
    } else {
        for (long i = threadIdx.x; i < c.count; i += 256) {
            float p = c.param[i], m = c.exp_avg[i], v = c.exp_avg_sq[i];
            upd(p, c.grad[i], m, v);
            c.param[i] = p; c.exp_avg[i] = m; c.exp_avg_sq[i] = v;
        }
    }
}

}  // namespace

extern "C" int dsc_grad_sumsq_f32(const dsc_optim_chunk* chunks, int32_t nchunks, double* partial, dsc_stream_t stream) {
    if (!chunks || !partial || nchunks < 1) return DSC_EINVAL;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(grad_sumsq_kernel, dim3(nchunks), dim3(256), 0, static_cast<hipStream_t>(stream), chunks, partial);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_clip_coef_f32(const double* partial, int32_t nchunks, float max_norm, float* total_norm,
                                 float* clip_coef, dsc_stream_t stream) {
    if (!partial || !total_norm || !clip_coef || nchunks < 1) return DSC_EINVAL;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), partial, nchunks,
                       max_norm, total_norm, clip_coef);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_adam_step_f32(const dsc_optim_chunk* chunks, int32_t nchunks, float step_size, float beta1, float beta2,
                                 float bias_correction2_sqrt, float eps, float weight_decay, const float* grad_scale,
                                 dsc_stream_t stream) {
    if (!chunks || nchunks < 1 || !(bias_correction2_sqrt > 0.f)) return DSC_EINVAL;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(adam_kernel, dim3(nchunks), dim3(256), 0, static_cast<hipStream_t>(stream), chunks, step_size, beta1,
                       beta2, bias_correction2_sqrt, eps, weight_decay, grad_scale);
    DSC_LAUNCH_CHECK();
    return 0;
}

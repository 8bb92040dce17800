// Elementwise DDPM steps (diffusion_ddpm.py:217-352,447-476).  HBM-bound: one read of each operand, one
// write of the result, coefficient gathers from device-resident tables.  Compiled with -ffp-contract=off:
// the reference evaluates  a*x + b*y  as two rounded products and a rounded sum; we do exactly that, so the
// results are bit-identical to the fp32 CPU path on the same inputs.
#define DSC_BAD_INDEX_COUNTER
#include "dsc_common.h"

namespace {

__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                      const int64_t* __restrict__ t, const float* __restrict__ sa,
                                                      const float* __restrict__ sb, float* __restrict__ xt,
                                                      float* __restrict__ vout, int64_t inner, int T) {
    const int b = blockIdx.y;
    const int64_t tv = dsc_checked_index(t[b], T, blockIdx.x == 0 && threadIdx.x == 0);      // one count per out-of-range scene
    const float a = sa[tv], s = sb[tv];
    const int64_t base = (int64_t)b * inner;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < inner; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = x0[base + i], n = noise[base + i];
        const float p0 = a * x, p1 = s * n;
        xt[base + i] = p0 + p1;
        if (vout) {
            const float q0 = a * n, q1 = s * x;
            vout[base + i] = q0 - q1;
        }
    }
}

__global__ __launch_bounds__(256) void p_sample_kernel(const float* xt, const float* __restrict__ mo,
                                                      const float* __restrict__ noise, const int64_t* __restrict__ t,
                                                      const float* __restrict__ ca, const float* __restrict__ cb,
                                                      const float* __restrict__ c1, const float* __restrict__ c2,
                                                      const float* __restrict__ sigma, float* out,   // out may alias xt (in-place step)
                                                      float* __restrict__ x0_out, int mean_type, int clip, int64_t inner, int T) {
    const int b = blockIdx.y;
    const int64_t tv = dsc_checked_index(t[b], T, blockIdx.x == 0 && threadIdx.x == 0);      // one count per out-of-range scene
    const float A = (mean_type == DSC_MEAN_X0) ? 0.f : ca[tv];
    const float Bc = (mean_type == DSC_MEAN_X0) ? 0.f : cb[tv];
    const float k1 = c1[tv], k2 = c2[tv];
    const float sg = (tv != 0) ? sigma[tv] : 0.f;
    const int64_t base = (int64_t)b * inner;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < inner; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = xt[base + i], m = mo[base + i];
        float x0;
        if (mean_type == DSC_MEAN_X0) x0 = m;
        else { const float p0 = A * x, p1 = Bc * m; x0 = p0 - p1; }
        if (clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        const float m0 = k1 * x0, m1 = k2 * x;
        const float mean = m0 + m1;
        const float nz = sg * noise[base + i];
        out[base + i] = mean + nz;
        if (x0_out) x0_out[base + i] = x0;
    }
}

__global__ void add_scalar_i64_kernel(int64_t* t, int count, int64_t delta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) t[i] += delta;
}

__global__ __launch_bounds__(256) void complete_overwrite_kernel(float* __restrict__ x, const float* __restrict__ partial,
                                                                const float* __restrict__ noise,
                                                                const int64_t* __restrict__ t, const float* __restrict__ sa,
                                                                const float* __restrict__ sb, int n, int p, int c, int T) {
    const int b = blockIdx.y;
    const int64_t tv = dsc_checked_index(t[b], T, blockIdx.x == 0 && threadIdx.x == 0);      // one count per out-of-range scene
    const float a = sa[tv], s = sb[tv];
    const int64_t cnt = (int64_t)p * c;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * blockDim.x) {
        const float p0 = a * partial[(int64_t)b * cnt + i], p1 = s * noise[(int64_t)b * cnt + i];
        x[(int64_t)b * n * c + i] = p0 + p1;   // rows [0,p) of scene b are the first p*c elements
    }
}

// Post-filter of generated scenes (reference delete_empty_from_network_samples, diffusion_scene_layout_ddpm.py:351-406): slot i
// of a scene is dropped when its 'empty' logit (column empty_col) is >= 0.  The reference takes that decision from BATCH ROW 0
// for every scene of the batch (:379, mode 0, kept as the drop-in default); mode 1 decides per scene, which is what batched
// generation needs.  One block per scene: keep flags -> exclusive prefix (ballot / popcount per wave, 3 waves cover N <= 192)
// -> kept rows move to the front in their original order, the tail is zero-filled, counts[b] = rows kept.
__global__ __launch_bounds__(192) void postfilter_compact_kernel(const float* __restrict__ x, int n, int c, int empty_col,
                                                                  int mode, int keep_empty, float* __restrict__ out,
                                                                  int* __restrict__ counts) {
    __shared__ int wave_cnt[3];
    __shared__ int dst[192];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* src_flags = x + (int64_t)(mode == 0 ? 0 : b) * n * c;
    const bool valid = tid < n;
    const bool keep = valid && (keep_empty || !(src_flags[(int64_t)tid * c + empty_col] >= 0.0f));
    const unsigned long long m = __ballot(keep);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wave_cnt[w];
    const int total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2];
    dst[tid] = keep ? base + before : -1;
    __syncthreads();
    const float* xs = x + (int64_t)b * n * c;
    float* os = out + (int64_t)b * n * c;
    for (int i = tid; i < n * c; i += 192) {
        const int r = i / c, col = i - r * c;
        const int d = dst[r];
        if (d >= 0) os[(int64_t)d * c + col] = xs[i];
    }
    for (int i = total * c + tid; i < n * c; i += 192) os[i] = 0.0f;
    if (tid == 0) counts[b] = total;
}

inline unsigned grid_x(int64_t inner) {
    int64_t g = (inner + 255) / 256;
    return (unsigned)(g > 64 ? 64 : g);
}

}  // namespace

extern "C" int dsc_q_sample_f32(const float* x0, const float* noise, const int64_t* t, const float* sqrt_ac,
                                const float* sqrt_1mac, float* x_t, float* v_out, int32_t b, int64_t inner,
                                int32_t num_timesteps, dsc_stream_t stream) {
    if (!x0 || !noise || !t || !sqrt_ac || !sqrt_1mac || !x_t || b < 1 || inner < 1 || num_timesteps < 1) return DSC_EINVAL;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(q_sample_kernel, dim3(grid_x(inner), b), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x0, noise, t, sqrt_ac, sqrt_1mac, x_t, v_out, inner, num_timesteps);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_p_sample_f32(const float* x_t, const float* model_out, const float* noise, const int64_t* t,
                                const float* ca, const float* cb, const float* coef1, const float* coef2,
                                const float* sigma, float* out, float* x0_out, int32_t mean_type, int32_t clip,
                                int32_t b, int64_t inner, int32_t num_timesteps, dsc_stream_t stream) {
    if (!x_t || !model_out || !noise || !t || !coef1 || !coef2 || !sigma || !out || b < 1 || inner < 1 || num_timesteps < 1)
        return DSC_EINVAL;
    if (mean_type < DSC_MEAN_EPS || mean_type > DSC_MEAN_V) return DSC_EINVAL;
    if (mean_type != DSC_MEAN_X0 && (!ca || !cb)) return DSC_EINVAL;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(p_sample_kernel, dim3(grid_x(inner), b), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x_t, model_out, noise, t, ca, cb, coef1, coef2, sigma, out, x0_out, mean_type, clip, inner, num_timesteps);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_add_scalar_i64(int64_t* t, int32_t count, int64_t delta, dsc_stream_t stream) {
    if (!t || count < 1) return DSC_EINVAL;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(add_scalar_i64_kernel, dim3((count + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), t, count, delta);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_postfilter_compact_f32(const float* samples, int32_t b, int32_t n, int32_t c, int32_t empty_col,
                                          int32_t mode, int32_t keep_empty, float* packed, int32_t* counts,
                                          dsc_stream_t stream) {
    if (!samples || !packed || !counts || b < 1 || n < 1 || c < 1 || empty_col < 0 || empty_col >= c) return DSC_EINVAL;
    if (mode < 0 || mode > 1 || samples == packed) return DSC_EINVAL;
    if (n > 192) return DSC_ERANGE;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(postfilter_compact_kernel, dim3(b), dim3(192), 0, static_cast<hipStream_t>(stream), samples, n, c,
                       empty_col, mode, keep_empty, packed, counts);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_complete_overwrite_f32(float* x, const float* partial, const float* noise, const int64_t* t,
                                          const float* sqrt_ac, const float* sqrt_1mac, int32_t b, int32_t n,
                                          int32_t p, int32_t c, int32_t num_timesteps, dsc_stream_t stream) {
    if (!x || !partial || !noise || !t || !sqrt_ac || !sqrt_1mac || b < 1 || n < 1 || p < 1 || p > n || c < 1 || num_timesteps < 1)
        return DSC_EINVAL;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(complete_overwrite_kernel, dim3(grid_x((int64_t)p * c), b), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, partial, noise, t, sqrt_ac, sqrt_1mac, n, p, c, num_timesteps);
    DSC_LAUNCH_CHECK();
    return 0;
}

unsigned dsc_bad_index_diffusion(bool reset) { return dsc_read_bad_index_count(reset); }

// Number of out-of-range device indices (timesteps outside [0, num_timesteps)) that kernels had to clamp since the
// last reset -- 0 in every correct run.  Synchronising (device-to-host reads of two counters): a test / debugging facility.
extern "C" int64_t dsc_device_error_count(int32_t reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    const unsigned a = dsc_bad_index_diffusion(reset != 0), b = dsc_bad_index_train(reset != 0);
    if (a == 0xffffffffu || b == 0xffffffffu) return -1;
    return (int64_t)a + b;
}

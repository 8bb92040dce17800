// Kernels of the FoldingNet KL auto-encoder that produces the 32/64-d shape codes of the scene denoiser (reference
// scene_synthesis/networks/foldingnet_autoencoder.py:56-390).  Point features are token-major
// [cloud * N + point][channel]; the 1x1 convolutions and Linear layers run on the fp32 MFMA GEMM (gemm_mfma.hip), this file
// holds what surrounds them:
//   knn_topk_kernel         k nearest neighbours of every point inside its cloud (reference knn(), :59-76), from xyz directly or
//                           from a Gram matrix computed by the batched GEMM
//   knn_cov_kernel          local covariance features of the encoder input (:197-205)
//   gather_max_kernel(+bwd) local max pooling over the k neighbours of a GraphLayer (:160-168)
//   bn_*                    BatchNorm1d in training mode (+ optional ReLU): column statistics over all points of the batch,
//                           apply, backward (column reductions of dy and dy * xhat, then the elementwise input gradient)
//   rowmax_kernel(+bwd)     global max pooling over the points of a cloud (:219)
//   point_affine_*          first convolution of a FoldingLayer (:247-251) without materialising cat([grid | codeword]):
//                           y[b, p, :] = Wp . x[b, p, 0:D] + t[b, :] with D = 2 (grid) or 3 (intermediate cloud) and
//                           t[b] = Wc . codeword[b] + bias computed once per cloud by the GEMM
#include "dsc_common.h"

namespace {

constexpr int KNN_K = 16;
constexpr int KNN_MAXC = 32;          // candidates per lane: clouds of up to 64 * 32 = 2048 points

// One wave per query point.  score_j = 2 <x_i, x_j> - |x_j|^2 (the reference ranks -|x_i|^2 + 2<x_i,x_j> - |x_j|^2; the
// first term is constant per query).  16 rounds of: per-lane maximum over its remaining candidates, wave arg-max (ties ->
// lowest index), winner retired.
template <bool XYZ>
__global__ __launch_bounds__(256) void knn_topk_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ gram,
                                                       const float* __restrict__ sq, int n, int clouds,
                                                       int* __restrict__ idx_out) {
    const int lane = threadIdx.x & 63;
    const long q = (long)blockIdx.x * 4 + (threadIdx.x >> 6);      // global query row
    if (q >= (long)clouds * n) return;
    const int b = (int)(q / n), i = (int)(q - (long)b * n);
    const long row0 = (long)b * n;
    float sc[KNN_MAXC];
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (XYZ) { qx = x[q * ldx]; qy = x[q * ldx + 1]; qz = x[q * ldx + 2]; }
#pragma unroll
    for (int c = 0; c < KNN_MAXC; ++c) {
        const int j = lane + 64 * c;
        float s = -INFINITY;
        if (j < n) {
            if (XYZ) {
                const float* pj = x + (row0 + j) * ldx;
                const float jx = pj[0], jy = pj[1], jz = pj[2];
                s = 2.0f * (qx * jx + qy * jy + qz * jz) - (jx * jx + jy * jy + jz * jz);
            } else {
                s = 2.0f * gram[((long)b * n + i) * n + j] - sq[row0 + j];
            }
        }
        sc[c] = s;
    }
    for (int r = 0; r < KNN_K; ++r) {
        float best = -INFINITY;
        int bj = 0x7fffffff;
#pragma unroll
        for (int c = 0; c < KNN_MAXC; ++c)
            if (sc[c] > best) { best = sc[c]; bj = lane + 64 * c; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oj = __shfl_xor(bj, o, 64);
            if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
        }
        if (lane == 0) idx_out[q * KNN_K + r] = bj;
        if ((bj & 63) == lane) {
            const int cc = bj >> 6;
#pragma unroll
            for (int c = 0; c < KNN_MAXC; ++c)
                if (c == cc) sc[c] = -INFINITY;
        }
    }
}

// |x_j|^2 per row (feature kNN)
__global__ void rowsq_kernel(const float* __restrict__ x, long ldx, int d, long rows, float* __restrict__ out) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float s = 0.f;
    for (int c = 0; c < d; ++c) { const float v = x[r * ldx + c]; s += v * v; }
    out[r] = s;
}

// out[q][0:3] = xyz, out[q][3:12] = sum_k (x_k - mean)(x_k - mean)^T over the 16 neighbours, row-major 3x3 (:199-205)
__global__ void knn_cov_kernel(const float* __restrict__ x, long ldx, const int* __restrict__ idx, int n, long rows,
                               float* __restrict__ out, long ldo) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= rows) return;
    const long row0 = (q / n) * n;
    float p[KNN_K][3];
    float m[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KNN_K; ++k) {
        const float* pj = x + (row0 + idx[q * KNN_K + k]) * ldx;
#pragma unroll
        for (int a = 0; a < 3; ++a) { p[k][a] = pj[a]; m[a] += pj[a]; }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) m[a] *= (1.0f / KNN_K);
    float cov[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KNN_K; ++k)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) cov[a * 3 + c] += (p[k][a] - m[a]) * (p[k][c] - m[c]);
#pragma unroll
    for (int a = 0; a < 3; ++a) out[q * ldo + a] = x[q * ldx + a];
#pragma unroll
    for (int a = 0; a < 9; ++a) out[q * ldo + 3 + a] = cov[a];
}

// out[q][c] = max_k x[nbr_k(q)][c], arg[q][c] = the k that won (first maximum, as torch.max)
__global__ __launch_bounds__(256) void gather_max_kernel(const float* __restrict__ x, long ldx, const int* __restrict__ idx,
                                                         int n, long rows, int ch, float* __restrict__ out, long ldo,
                                                         unsigned char* __restrict__ arg) {
    const int lane = threadIdx.x & 63;
    const long q = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= rows) return;
    const long row0 = (q / n) * n;
    int nb[KNN_K];
#pragma unroll
    for (int k = 0; k < KNN_K; ++k) nb[k] = idx[q * KNN_K + k];
    for (int c = lane; c < ch; c += 64) {
        float best = x[(row0 + nb[0]) * ldx + c];
        int bk = 0;
#pragma unroll
        for (int k = 1; k < KNN_K; ++k) {
            const float v = x[(row0 + nb[k]) * ldx + c];
            if (v > best) { best = v; bk = k; }
        }
        out[q * ldo + c] = best;
        arg[q * ch + c] = (unsigned char)bk;
    }
}

// dx[nbr_arg(q,c)][c] += dy[q][c]   (dx zero-filled by the caller; fp32 atomics: the order of the additions is not fixed)
__global__ __launch_bounds__(256) void gather_max_bwd_kernel(const float* __restrict__ dy, long ldy,
                                                             const int* __restrict__ idx, const unsigned char* __restrict__ arg,
                                                             int n, long rows, int ch, float* __restrict__ dx, long lddx) {
    const int lane = threadIdx.x & 63;
    const long q = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= rows) return;
    const long row0 = (q / n) * n;
    for (int c = lane; c < ch; c += 64) {
        const int src = idx[q * KNN_K + arg[q * ch + c]];
        atomicAdd(dx + (row0 + src) * lddx + c, dy[q * ldy + c]);
    }
}

// ---- column statistics: partial[split][0..1][C] = sum (v - pivot), sum (v - pivot)^2 over the rows of the split, for
// v = a (MODE 0) | sums of g and g * b with g = a masked by (mask > 0) (MODE 1: BatchNorm backward: a = dy, b = xhat,
// mask = y of a ReLU layer or nullptr)
template <int MODE>
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b,
                                                       long ldb, const float* __restrict__ mask, long ldm,
                                                       const float* __restrict__ pivot, long rows, int ch, int chunk,
                                                       float* __restrict__ partial) {
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long r0 = (long)blockIdx.y * chunk;
    const long r1 = (r0 + chunk < rows) ? r0 + chunk : rows;
    float s0 = 0.f, s1 = 0.f;
    if (c < ch) {
        const float pv = (MODE == 0 && pivot) ? pivot[c] : 0.f;
        for (long r = r0 + rl; r < r1; r += 4) {
            if (MODE == 0) {
                const float v = a[r * lda + c] - pv;
                s0 += v; s1 += v * v;
            } else {
                float g = a[r * lda + c];
                if (mask && !(mask[r * ldm + c] > 0.f)) g = 0.f;
                s0 += g; s1 += g * b[r * ldb + c];
            }
        }
    }
    red[0][rl][cl] = s0; red[1][rl][cl] = s1;
    __syncthreads();
    if (rl == 0 && c < ch) {
        partial[((long)blockIdx.y * 2 + 0) * ch + c] = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
        partial[((long)blockIdx.y * 2 + 1) * ch + c] = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    }
}

// BatchNorm finalize: mean / rstd from the partial sums (fp64 combine), running statistics (momentum, unbiased variance)
__global__ void bn_finalize_kernel(const float* __restrict__ partial, int splits, const float* __restrict__ pivot, long rows,
                                   int ch, float eps, float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                   float* __restrict__ run_mean, float* __restrict__ run_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ch) return;
    double s0 = 0.0, s1 = 0.0;
    for (int s = 0; s < splits; ++s) { s0 += partial[((long)s * 2) * ch + c]; s1 += partial[((long)s * 2 + 1) * ch + c]; }
    const double m = s0 / (double)rows;
    double var = s1 / (double)rows - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)(m + (double)pivot[c]);
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
        const double unb = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
        run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mean[c];
        run_var[c] = (1.0f - momentum) * run_var[c] + momentum * (float)unb;
    }
}

// xhat = (x - mean) * rstd;  y = relu?(xhat * gamma + beta)
__global__ void bn_apply_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ mean,
                                const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                long rows, int ch, int relu, float* __restrict__ xhat, float* __restrict__ y) {
    const long total = rows * ch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / ch;
        const int c = (int)(i - r * ch);
        const float xh = (x[r * ldx + c] - mean[c]) * rstd[c];
        float v = xh * gamma[c] + beta[c];
        if (relu && v < 0.f) v = 0.f;
        if (xhat) xhat[i] = xh;
        y[i] = v;
    }
}

// dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy masked by the ReLU;  dgamma = sum g * xhat, dbeta = sum g
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ xhat, const float* __restrict__ y,
                                    const float* __restrict__ gamma, const float* __restrict__ rstd,
                                    const float* __restrict__ sums /* [2][C]: sum g, sum g*xhat */, long rows, int ch, int relu,
                                    float* __restrict__ dx) {
    const long total = rows * ch;
    const float inv = 1.0f / (float)rows;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % ch);
        float g = dy[i];
        if (relu && !(y[i] > 0.f)) g = 0.f;
        dx[i] = gamma[c] * rstd[c] * (g - sums[c] * inv - xhat[i] * (sums[ch + c] * inv));
    }
}

// out[s][c] = sum over splits of partial[split][s][c]  (s = 0, 1)
__global__ void colstats_reduce_kernel(const float* __restrict__ partial, int splits, int ch, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * ch) return;
    const int s = i / ch, c = i - s * ch;
    double acc = 0.0;
    for (int k = 0; k < splits; ++k) acc += partial[((long)k * 2 + s) * ch + c];
    out[i] = (float)acc;
}

// global max pooling over the points of a cloud: out[b][c] = max_p x[b*n + p][c], arg = first p that attains it
__global__ __launch_bounds__(256) void rowmax_kernel(const float* __restrict__ x, long ldx, int n, int ch,
                                                     float* __restrict__ out, int* __restrict__ arg) {
    __shared__ float bv[4][64];
    __shared__ int bi[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, b = blockIdx.y;
    float best = -INFINITY;
    int bp = 0x7fffffff;
    if (c < ch)
        for (int p = rl; p < n; p += 4) {
            const float v = x[((long)b * n + p) * ldx + c];
            if (v > best) { best = v; bp = p; }
        }
    bv[rl][cl] = best; bi[rl][cl] = bp;
    __syncthreads();
    if (rl == 0 && c < ch) {
        for (int k = 1; k < 4; ++k)
            if (bv[k][cl] > best || (bv[k][cl] == best && bi[k][cl] < bp)) { best = bv[k][cl]; bp = bi[k][cl]; }
        out[(long)b * ch + c] = best;
        arg[(long)b * ch + c] = bp;
    }
}

__global__ void rowmax_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ arg, int n, int ch, long rows,
                                  float* __restrict__ dx) {
    const long total = rows * ch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / ch;
        const int c = (int)(i - r * ch);
        const long b = r / n;
        dx[i] = (arg[b * ch + c] == (int)(r - b * n)) ? dy[b * ch + c] : 0.f;
    }
}

// y[b*n + p][c] = sum_{d < D} wp[c][d] * x[xrow][d] + t[b][c];  xrow = p (shared grid, x_per_cloud == 0) or b*n + p
__global__ void point_affine_kernel(const float* __restrict__ x, long ldx, int x_per_cloud, const float* __restrict__ wp,
                                    long ldw, const float* __restrict__ t, int n, int ch, int d, long rows,
                                    float* __restrict__ y) {
    const long total = rows * ch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / ch;
        const int c = (int)(i - r * ch);
        const long b = r / n;
        const float* xr = x + (x_per_cloud ? r : r - b * n) * ldx;
        float v = t[b * ch + c];
        for (int k = 0; k < d; ++k) v += wp[c * ldw + k] * xr[k];
        y[i] = v;
    }
}

// dx[r][k] = sum_c dy[r][c] * wp[c][k]   (one wave per row, D <= 4)
__global__ __launch_bounds__(256) void point_affine_dx_kernel(const float* __restrict__ dy, const float* __restrict__ wp,
                                                              long ldw, int ch, int d, long rows, float* __restrict__ dx,
                                                              long lddx) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = lane; c < ch; c += 64) {
        const float g = dy[r * ch + c];
        for (int k = 0; k < d; ++k) acc[k] += g * wp[c * ldw + k];
    }
    for (int k = 0; k < d; ++k) {
        const float s = wave_sum(acc[k]);
        if (lane == 0) dx[r * lddx + k] = s;
    }
}

// partial[split][k][c] = sum over the split's rows of dy[r][c] * x[xrow][k]   (k < D <= 4): the D weight-gradient columns
__global__ __launch_bounds__(256) void point_affine_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              long ldx, int x_per_cloud, int n, int ch, int d, long rows,
                                                              int chunk, float* __restrict__ partial) {
    __shared__ float red[4][4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long r0 = (long)blockIdx.y * chunk;
    const long r1 = (r0 + chunk < rows) ? r0 + chunk : rows;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < ch)
        for (long r = r0 + rl; r < r1; r += 4) {
            const float g = dy[r * ch + c];
            const float* xr = x + (x_per_cloud ? r : r % n) * ldx;
            for (int k = 0; k < d; ++k) acc[k] += g * xr[k];
        }
    for (int k = 0; k < 4; ++k) red[k][rl][cl] = acc[k];
    __syncthreads();
    if (rl == 0 && c < ch)
        for (int k = 0; k < d; ++k)
            partial[((long)blockIdx.y * 4 + k) * ch + c] = (red[k][0][cl] + red[k][1][cl]) + (red[k][2][cl] + red[k][3][cl]);
}

__global__ void point_affine_dw_reduce_kernel(const float* __restrict__ partial, int splits, int ch, int d,
                                              float* __restrict__ dw, long lddw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d * ch) return;
    const int k = i / ch, c = i - k * ch;
    double acc = 0.0;
    for (int s = 0; s < splits; ++s) acc += partial[((long)s * 4 + k) * ch + c];
    dw[c * lddw + k] = (float)acc;
}

// dt[b][c] = sum_p dy[b*n + p][c]
__global__ __launch_bounds__(256) void cloud_colsum_kernel(const float* __restrict__ dy, int n, int ch, float* __restrict__ dt) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, b = blockIdx.y;
    float s = 0.f;
    if (c < ch)
        for (int p = rl; p < n; p += 4) s += dy[((long)b * n + p) * ch + c];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < ch) dt[(long)b * ch + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

inline unsigned ew_grid(long total) {
    long g = (total + 255) / 256;
    return (unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}
inline int split_rows(long rows, int* chunk) {
    int splits = (int)((rows + 511) / 512);
    if (splits > 256) splits = 256;
    if (splits < 1) splits = 1;
    *chunk = (int)((rows + splits - 1) / splits);
    return (int)((rows + *chunk - 1) / *chunk);
}

}  // namespace

extern "C" int dsc_knn16_f32(const float* x, int64_t ldx, int32_t dim, const float* gram, const float* sqnorm, int32_t clouds,
                             int32_t n, int32_t* idx, dsc_stream_t stream) {
    if (!x || !idx || clouds < 1 || n < KNN_K) return DSC_EINVAL;
    if (n > 64 * KNN_MAXC) return DSC_ERANGE;
    if (dim != 3 && (!gram || !sqnorm)) return DSC_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long rows = (long)clouds * n;
    DSC_CLEAR_STALE_ERROR();
    if (dim == 3 && !gram)
        hipLaunchKernelGGL(knn_topk_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, (long)ldx, nullptr, nullptr,
                           n, clouds, idx);
    else
        hipLaunchKernelGGL(knn_topk_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, (long)ldx, gram, sqnorm,
                           n, clouds, idx);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_rowsq_f32(const float* x, int64_t ldx, int32_t dim, int64_t rows, float* out, dsc_stream_t stream) {
    if (!x || !out || dim < 1 || rows < 1) return DSC_EINVAL;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(rowsq_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                       (long)ldx, dim, (long)rows, out);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_knn_cov_f32(const float* xyz, int64_t ldx, const int32_t* idx, int32_t clouds, int32_t n, float* out,
                               int64_t ldo, dsc_stream_t stream) {
    if (!xyz || !idx || !out || clouds < 1 || n < KNN_K || ldo < 12) return DSC_EINVAL;
    const long rows = (long)clouds * n;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(knn_cov_kernel, dim3((unsigned)((rows + 127) / 128)), dim3(128), 0, static_cast<hipStream_t>(stream), xyz,
                       (long)ldx, idx, n, rows, out, (long)ldo);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_gather_max_f32(const float* x, int64_t ldx, const int32_t* idx, int32_t clouds, int32_t n, int32_t ch,
                                  float* out, int64_t ldo, uint8_t* arg, dsc_stream_t stream) {
    if (!x || !idx || !out || !arg || clouds < 1 || n < 1 || ch < 1) return DSC_EINVAL;
    const long rows = (long)clouds * n;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(gather_max_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                       (long)ldx, idx, n, rows, ch, out, (long)ldo, arg);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_gather_max_bwd_f32(const float* dy, int64_t ldy, const int32_t* idx, const uint8_t* arg, int32_t clouds,
                                      int32_t n, int32_t ch, float* dx_zeroed, int64_t lddx, dsc_stream_t stream) {
    if (!dy || !idx || !arg || !dx_zeroed || clouds < 1 || n < 1 || ch < 1) return DSC_EINVAL;
    const long rows = (long)clouds * n;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(gather_max_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       dy, (long)ldy, idx, arg, n, rows, ch, dx_zeroed, (long)lddx);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t dsc_bn_workspace_floats(int64_t rows, int32_t ch) {
    int chunk;
    return (int64_t)split_rows(rows, &chunk) * 4 * ch;
}

extern "C" int dsc_batchnorm_fwd_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, int64_t rows,
                                     int32_t ch, float eps, float momentum, int32_t relu, float* mean, float* rstd,
                                     float* running_mean, float* running_var, float* xhat, float* y, float* workspace,
                                     int64_t workspace_floats, dsc_stream_t stream) {
    if (!x || !gamma || !beta || !mean || !rstd || !y || !workspace || rows < 1 || ch < 1) return DSC_EINVAL;
    int chunk;
    const int splits = split_rows(rows, &chunk);
    if (workspace_floats < (int64_t)splits * 4 * ch) return DSC_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    DSC_CLEAR_STALE_ERROR();
    // pivot = first row (shifted sums: no cancellation in the variance)
    hipLaunchKernelGGL(colstats_kernel<0>, dim3((ch + 63) / 64, splits), dim3(256), 0, s, x, (long)ldx, nullptr, 0L, nullptr, 0L,
                       x, (long)rows, ch, chunk, workspace);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((ch + 255) / 256), dim3(256), 0, s, workspace, splits, x, (long)rows, ch, eps,
                       momentum, mean, rstd, running_mean, running_var);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(rows * ch)), dim3(256), 0, s, x, (long)ldx, mean, rstd, gamma, beta,
                       (long)rows, ch, relu, xhat, y);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_batchnorm_eval_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, const float* mean,
                                      const float* rstd, int64_t rows, int32_t ch, int32_t relu, float* y, dsc_stream_t stream) {
    if (!x || !gamma || !beta || !mean || !rstd || !y || rows < 1 || ch < 1) return DSC_EINVAL;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(rows * ch)), dim3(256), 0, static_cast<hipStream_t>(stream), x, (long)ldx,
                       mean, rstd, gamma, beta, (long)rows, ch, relu, nullptr, y);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_batchnorm_bwd_f32(const float* dy, const float* xhat, const float* y, const float* gamma, const float* rstd,
                                     int64_t rows, int32_t ch, int32_t relu, float* dx, float* dgamma, float* dbeta,
                                     float* workspace, int64_t workspace_floats, dsc_stream_t stream) {
    if (!dy || !xhat || !gamma || !rstd || !dx || !dgamma || !dbeta || !workspace || rows < 1 || ch < 1) return DSC_EINVAL;
    if (relu && !y) return DSC_EINVAL;
    int chunk;
    const int splits = split_rows(rows, &chunk);
    if (workspace_floats < (int64_t)splits * 4 * ch) return DSC_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* sums = workspace + (int64_t)splits * 2 * ch;          // [2][C] behind the partials
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(colstats_kernel<1>, dim3((ch + 63) / 64, splits), dim3(256), 0, s, dy, (long)ch, xhat, (long)ch,
                       relu ? y : nullptr, (long)ch, nullptr, (long)rows, ch, chunk, workspace);
    hipLaunchKernelGGL(colstats_reduce_kernel, dim3((2 * ch + 255) / 256), dim3(256), 0, s, workspace, splits, ch, sums);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(rows * ch)), dim3(256), 0, s, dy, xhat, y, gamma, rstd, sums, (long)rows,
                       ch, relu, dx);
    // dbeta = sum g, dgamma = sum g * xhat
    DSC_LAUNCH_CHECK();
    hipError_t e = hipMemcpyAsync(dbeta, sums, sizeof(float) * ch, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dgamma, sums + ch, sizeof(float) * ch, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return (int)e;
    return 0;
}

extern "C" int dsc_rowmax_f32(const float* x, int64_t ldx, int32_t clouds, int32_t n, int32_t ch, float* out, int32_t* arg,
                              dsc_stream_t stream) {
    if (!x || !out || !arg || clouds < 1 || n < 1 || ch < 1) return DSC_EINVAL;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(rowmax_kernel, dim3((ch + 63) / 64, clouds), dim3(256), 0, static_cast<hipStream_t>(stream), x, (long)ldx,
                       n, ch, out, arg);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_rowmax_bwd_f32(const float* dy, const int32_t* arg, int32_t clouds, int32_t n, int32_t ch, float* dx,
                                  dsc_stream_t stream) {
    if (!dy || !arg || !dx || clouds < 1 || n < 1 || ch < 1) return DSC_EINVAL;
    const long rows = (long)clouds * n;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(rowmax_bwd_kernel, dim3(ew_grid(rows * ch)), dim3(256), 0, static_cast<hipStream_t>(stream), dy, arg, n, ch,
                       rows, dx);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_point_affine_f32(const float* x, int64_t ldx, int32_t x_per_cloud, const float* wp, int64_t ldw,
                                    const float* t, int32_t clouds, int32_t n, int32_t ch, int32_t d, float* y,
                                    dsc_stream_t stream) {
    if (!x || !wp || !t || !y || clouds < 1 || n < 1 || ch < 1 || d < 1 || d > 4) return DSC_EINVAL;
    const long rows = (long)clouds * n;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(point_affine_kernel, dim3(ew_grid(rows * ch)), dim3(256), 0, static_cast<hipStream_t>(stream), x, (long)ldx,
                       x_per_cloud, wp, (long)ldw, t, n, ch, d, rows, y);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_point_affine_bwd_f32(const float* dy, const float* x, int64_t ldx, int32_t x_per_cloud, const float* wp,
                                        int64_t ldw, int32_t clouds, int32_t n, int32_t ch, int32_t d, float* dx /* may be NULL */,
                                        int64_t lddx, float* dwp, int64_t lddw, float* dt, float* workspace,
                                        int64_t workspace_floats, dsc_stream_t stream) {
    if (!dy || !x || !wp || !dwp || !dt || !workspace || clouds < 1 || n < 1 || ch < 1 || d < 1 || d > 4) return DSC_EINVAL;
    const long rows = (long)clouds * n;
    int chunk;
    const int splits = split_rows(rows, &chunk);
    if (workspace_floats < (int64_t)splits * 4 * ch) return DSC_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    DSC_CLEAR_STALE_ERROR();
    if (dx)
        hipLaunchKernelGGL(point_affine_dx_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, dy, wp, (long)ldw, ch, d, rows,
                           dx, (long)lddx);
    hipLaunchKernelGGL(point_affine_dw_kernel, dim3((ch + 63) / 64, splits), dim3(256), 0, s, dy, x, (long)ldx, x_per_cloud, n, ch,
                       d, rows, chunk, workspace);
    hipLaunchKernelGGL(point_affine_dw_reduce_kernel, dim3((d * ch + 255) / 256), dim3(256), 0, s, workspace, splits, ch, d, dwp,
                       (long)lddw);
    hipLaunchKernelGGL(cloud_colsum_kernel, dim3((ch + 63) / 64, clouds), dim3(256), 0, s, dy, n, ch, dt);
    DSC_LAUNCH_CHECK();
    return 0;
}

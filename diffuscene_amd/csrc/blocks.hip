// Non-GEMM denoiser kernels: weight standardisation, channel LayerNorm, linear / softmax attention cores,
// small-K input projection, time embedding, activation.  All HBM- or latency-bound; one pass over the data,
// wave-level shuffles for the reductions, K/V tiles of one scene staged in LDS.
#include "dsc_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// WeightStandardizedConv2d weight path (denoise_net.py:84-89): one 256-thread block per weight row.
// ------------------------------------------------------------------------------------------------
struct WsBatch { dsc_ws_item it[DSC_WS_MAX]; };

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void ws_kernel(const WsBatch b, const float eps) {
    __shared__ float red[4];
    const dsc_ws_item it = b.it[blockIdx.y];
    const int row = blockIdx.x;
    if (row >= it.rows) return;
    const float* w = it.w + (int64_t)row * it.cols;
    float* o = it.out + (int64_t)row * it.cols;
    constexpr int MAXE = 8;   // cols <= 2048
    float v[MAXE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int c = threadIdx.x + 256 * i;
        v[i] = (c < it.cols) ? w[c] : 0.f;
        s += v[i];
    }
    const float mean = block_sum_256(s, red) / (float)it.cols;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int c = threadIdx.x + 256 * i;
        const float d = (c < it.cols) ? v[i] - mean : 0.f;
        s2 += d * d;
    }
    const float var = block_sum_256(s2, red) / (float)it.cols;
    const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < it.cols) o[c] = (v[i] - mean) * rs;
    }
}

// ------------------------------------------------------------------------------------------------
// channel LayerNorm with gain (+ residual), d = 512: one wave per token row, 8 floats per lane.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm512_kernel(const float* __restrict__ x, int64_t ldx,
                                                          const float* __restrict__ g,
                                                          const float* __restrict__ res, int64_t ldr,
                                                          float* __restrict__ y, int64_t ldy, int m, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    const float* xr = x + (int64_t)row * ldx;
    const f32x4 a = *reinterpret_cast<const f32x4*>(xr + lane * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(xr + 256 + lane * 4);
    float s = (a[0] + a[1]) + (a[2] + a[3]) + (b[0] + b[1]) + (b[2] + b[3]);
    const float mean = wave_sum(s) * (1.0f / 512.0f);
    float s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float da = a[e] - mean, db = b[e] - mean;
        s2 += da * da + db * db;
    }
    const float var = wave_sum(s2) * (1.0f / 512.0f);
    const float rs = 1.0f / sqrtf(var + eps);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(g + lane * 4);
    const f32x4 gb = *reinterpret_cast<const f32x4*>(g + 256 + lane * 4);
    f32x4 oa, ob;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        oa[e] = (a[e] - mean) * rs * ga[e];
        ob[e] = (b[e] - mean) * rs * gb[e];
    }
    if (res) {
        const float* rr = res + (int64_t)row * ldr;
        const f32x4 ra = *reinterpret_cast<const f32x4*>(rr + lane * 4);
        const f32x4 rb = *reinterpret_cast<const f32x4*>(rr + 256 + lane * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { oa[e] += ra[e]; ob[e] += rb[e]; }
    }
    float* yr = y + (int64_t)row * ldy;
    *reinterpret_cast<f32x4*>(yr + lane * 4) = oa;
    *reinterpret_cast<f32x4*>(yr + 256 + lane * 4) = ob;
}

// ------------------------------------------------------------------------------------------------
// Linear attention core, one 256-thread block per (scene, head); Q/K/V head slices of the scene in LDS
// (rows padded to 36 floats: 16-byte aligned float4 rows, column walks spread over the banks).
//   phase 1  k <- softmax over the tokens        8 threads per head channel
//   phase 2  ctx[d][e] = sum_j k[j][d] v[j][e]   4 outputs per thread
//   phase 3  q <- softmax over channels * scale; out[i][e] = sum_d ctx[d][e] q[i][d]   4 threads per token
// ------------------------------------------------------------------------------------------------
constexpr int MAXTOK = 160;
constexpr int HP = 33;      // softmax-attention kernel below
constexpr int LP = 36;

__global__ __launch_bounds__(256) void linear_attention_kernel(const float* __restrict__ q, int64_t ldq,
                                                               const float* __restrict__ k, int64_t ldk,
                                                               const float* __restrict__ v, int64_t ldv,
                                                               float* __restrict__ out, int64_t ldo,
                                                               int nq, int nk, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ks = lds;                      // [nk][LP]
    float* Vs = Ks + nk * LP;             // [nk][LP]
    float* Qs = Vs + nk * LP;             // [nq][LP]
    float* ctx = Qs + nq * LP;            // [32][LP]
    const int b = blockIdx.x >> 2, h = blockIdx.x & 3;
    const int tid = threadIdx.x;
    const float* kb = k + (int64_t)b * nk * ldk + h * 32;
    const float* vb = v + (int64_t)b * nk * ldv + h * 32;
    const float* qb = q + (int64_t)b * nq * ldq + h * 32;
    for (int f = tid; f < nk * 8; f += 256) {
        const int j = f >> 3, c4 = (f & 7) * 4;
        *reinterpret_cast<f32x4*>(Ks + j * LP + c4) = *reinterpret_cast<const f32x4*>(kb + (int64_t)j * ldk + c4);
        *reinterpret_cast<f32x4*>(Vs + j * LP + c4) = *reinterpret_cast<const f32x4*>(vb + (int64_t)j * ldv + c4);
    }
    for (int f = tid; f < nq * 8; f += 256) {
        const int i = f >> 3, c4 = (f & 7) * 4;
        *reinterpret_cast<f32x4*>(Qs + i * LP + c4) = *reinterpret_cast<const f32x4*>(qb + (int64_t)i * ldq + c4);
    }
    __syncthreads();
    {   // phase 1: softmax of k over the nk tokens; 8 consecutive lanes share a channel
        const int d = tid >> 3, part = tid & 7;
        float mx = -INFINITY;
        for (int j = part; j < nk; j += 8) mx = fmaxf(mx, Ks[j * LP + d]);
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
        float sm = 0.f;
        for (int j = part; j < nk; j += 8) {
            const float e = expf(Ks[j * LP + d] - mx);
            Ks[j * LP + d] = e;
            sm += e;
        }
        sm += __shfl_xor(sm, 1, 64);
        sm += __shfl_xor(sm, 2, 64);
        sm += __shfl_xor(sm, 4, 64);
        const float inv = 1.0f / sm;
        for (int j = part; j < nk; j += 8) Ks[j * LP + d] *= inv;
    }
    __syncthreads();
    {   // phase 2: context; thread -> (d, 4 consecutive e)
        const int d = tid >> 3, e0 = (tid & 7) * 4;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < nk; ++j) {
            const float kd = Ks[j * LP + d];
            const f32x4 vv = *reinterpret_cast<const f32x4*>(Vs + j * LP + e0);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += kd * vv[e];
        }
        *reinterpret_cast<f32x4*>(ctx + d * LP + e0) = a;
    }
    __syncthreads();
    // phase 3: 4 lanes per query token, each owns 8 channels / 8 outputs
    for (int i0 = 0; i0 < nq; i0 += 64) {
        const int i = i0 + (tid >> 2), part = tid & 3;
        const bool ok = i < nq;
        float qv[8];
        float mx = -INFINITY;
        if (ok) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(Qs + i * LP + part * 8);
            const f32x4 c = *reinterpret_cast<const f32x4*>(Qs + i * LP + part * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { qv[e] = a[e]; qv[4 + e] = c[e]; }
#pragma unroll
            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, qv[e]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        float sm = 0.f;
        if (ok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { qv[e] = expf(qv[e] - mx); sm += qv[e]; }
        }
        sm += __shfl_xor(sm, 1, 64);
        sm += __shfl_xor(sm, 2, 64);
        if (ok) {
            const float inv = scale / sm;
            f32x4 a, c;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = qv[e] * inv; c[e] = qv[4 + e] * inv; }
            *reinterpret_cast<f32x4*>(Qs + i * LP + part * 8) = a;         // row i is touched by its own 4 lanes only
            *reinterpret_cast<f32x4*>(Qs + i * LP + part * 8 + 4) = c;
        }
        __builtin_amdgcn_wave_barrier();
        if (ok) {
            f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int d = 0; d < 32; ++d) {
                const float qd = Qs[i * LP + d];
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(ctx + d * LP + part * 8);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(ctx + d * LP + part * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { o0[e] += c0[e] * qd; o1[e] += c1[e] * qd; }
            }
            float* orow = out + ((int64_t)b * nq + i) * ldo + h * 32 + part * 8;
            *reinterpret_cast<f32x4*>(orow) = o0;
            *reinterpret_cast<f32x4*>(orow + 4) = o1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Softmax attention core (mid_attn), one block of 256 threads per (scene, head).  K/V of the scene are staged in LDS
// (36-float rows, 16-byte reads); the block has as many waves as the scene has 16-token groups (dsc_attention_threads: 320 threads at
// N = 80 -- one pass over the queries instead of two with the second three quarters idle; 75 -> 4x us per launch).
// FOUR lanes share a query token, each owning 8 of the 32 head channels: a score is
// 8 FMAs + two wave shuffles across the quad, every lane of a quad walks the same key (LDS broadcast).  Two passes over
// the keys (row maximum, then exp / sum / PV), i.e. the plain softmax of the reference (denoise_net.py:252-258).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void attention_kernel(const float* __restrict__ q, int64_t ldq,
                                                        const float* __restrict__ k, int64_t ldk,
                                                        const float* __restrict__ v, int64_t ldv,
                                                        float* __restrict__ out, int64_t ldo, int n, float scale) {
    extern __shared__ __attribute__((aligned(16))) float att_lds[];
    float* Ks = att_lds;                  // [n][LP]
    float* Vs = Ks + n * LP;              // [n][LP]
    const int b = blockIdx.x >> 2, h = blockIdx.x & 3;
    const int tid = threadIdx.x;
    const float* kb = k + (int64_t)b * n * ldk + h * 32;
    const float* vb = v + (int64_t)b * n * ldv + h * 32;
    const int nthr = blockDim.x, rows = nthr >> 2;       // four lanes per query token (dsc_attention_threads)
    for (int f = tid; f < n * 8; f += nthr) {
        const int j = f >> 3, c4 = (f & 7) * 4;
        *reinterpret_cast<f32x4*>(Ks + j * LP + c4) = *reinterpret_cast<const f32x4*>(kb + (int64_t)j * ldk + c4);
        *reinterpret_cast<f32x4*>(Vs + j * LP + c4) = *reinterpret_cast<const f32x4*>(vb + (int64_t)j * ldv + c4);
    }
    __syncthreads();
    const int part = tid & 3;
    for (int i0 = 0; i0 < n; i0 += rows) {
        const int i = i0 + (tid >> 2);
        const bool ok = i < n;
        const float* qr = q + ((int64_t)b * n + (ok ? i : 0)) * ldq + h * 32 + part * 8;
        const f32x4 qa = *reinterpret_cast<const f32x4*>(qr), qc = *reinterpret_cast<const f32x4*>(qr + 4);
        float qv[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { qv[e] = qa[e] * scale; qv[4 + e] = qc[e] * scale; }
        auto score = [&](int j) {
            const f32x4 ka = *reinterpret_cast<const f32x4*>(Ks + j * LP + part * 8);
            const f32x4 kc = *reinterpret_cast<const f32x4*>(Ks + j * LP + part * 8 + 4);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) s += qv[e] * ka[e];
#pragma unroll
            for (int e = 0; e < 4; ++e) s += qv[4 + e] * kc[e];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            return s;
        };
        float mx = -INFINITY;
        for (int j = 0; j < n; ++j) mx = fmaxf(mx, score(j));
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float sm = 0.f;
        for (int j = 0; j < n; ++j) {
            const float pj = expf(score(j) - mx);
            sm += pj;
            const f32x4 va = *reinterpret_cast<const f32x4*>(Vs + j * LP + part * 8);
            const f32x4 vc = *reinterpret_cast<const f32x4*>(Vs + j * LP + part * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] += pj * va[e]; o[4 + e] += pj * vc[e]; }
        }
        if (ok) {
            const float inv = 1.0f / sm;
            float* orow = out + ((int64_t)b * n + i) * ldo + h * 32 + part * 8;
            f32x4 t0, t1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { t0[e] = o[e] * inv; t1[e] = o[4 + e] * inv; }
            *reinterpret_cast<f32x4*>(orow) = t0;
            *reinterpret_cast<f32x4*>(orow + 4) = t1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// small-K linear on un-aligned rows: 32 tokens per block staged in LDS as [k][token]; thread = channel.
// ------------------------------------------------------------------------------------------------
constexpr int SK_TOK = 32;
constexpr int SK_MAXK = 64;

__device__ __forceinline__ void linear_smallk_block(const float* __restrict__ x, int64_t ldx, int kin,
                                                    const float* __restrict__ w, int64_t ldw,
                                                    const float* __restrict__ bias,
                                                    float* __restrict__ y, int64_t ldy, int m, int n, int act, float* xs) {
    const int tok0 = blockIdx.x * SK_TOK;
    const int ntok = (m - tok0) < SK_TOK ? (m - tok0) : SK_TOK;
    for (int f = threadIdx.x; f < SK_TOK * kin; f += 256) {
        const int t = f / kin, kk = f - t * kin;
        xs[kk * SK_TOK + t] = (t < ntok) ? x[(int64_t)(tok0 + t) * ldx + kk] : 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < n; c += 256) {
        float acc[SK_TOK];
        const float b = bias ? bias[c] : 0.f;
#pragma unroll
        for (int t = 0; t < SK_TOK; ++t) acc[t] = b;
        const float* wr = w + (int64_t)c * ldw;
        // 8 weights in flight per thread: one L2 round trip per 8 k instead of one per k (the serialised loads were the whole
        // 30 us of this launch)
        for (int k0 = 0; k0 < kin; k0 += 8) {
            float wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[j] = (k0 + j < kin) ? wr[k0 + j] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (k0 + j < kin) {
#pragma unroll
                    for (int t4 = 0; t4 < SK_TOK / 4; ++t4) {
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(&xs[(k0 + j) * SK_TOK + t4 * 4]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[t4 * 4 + e] += wv[j] * xv[e];
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < SK_TOK; ++t)
            if (t < ntok) y[(int64_t)(tok0 + t) * ldy + c] = dsc_act(acc[t], act);
    }
}

__global__ __launch_bounds__(256) void linear_smallk_kernel(const float* __restrict__ x, int64_t ldx, int kin,
                                                            const float* __restrict__ w, int64_t ldw,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ y, int64_t ldy, int m, int n, int act) {
    __shared__ float xs[SK_MAXK * SK_TOK];
    linear_smallk_block(x, ldx, kin, w, ldw, bias, y, ldy, m, n, act, xs);
}

// the first layers of ALL per-attribute encoders in one launch (blockIdx.y = head): 3-4 x the blocks of a single head in flight
struct SmallKBatch { dsc_smallk_item it[DSC_SMALLK_MAX]; };
__global__ __launch_bounds__(256) void linear_smallk_grouped_kernel(const SmallKBatch b, int m, int n, int act) {
    __shared__ float xs[SK_MAXK * SK_TOK];
    const dsc_smallk_item it = b.it[blockIdx.y];
    linear_smallk_block(it.x, it.ldx, it.k_in, it.w, it.ldw, it.bias, it.y, it.ldy, m, n, act, xs);
}

// dst[r][span.dst_col + c] = src[r][span.src_col + c], c < span.width, for a few column spans (the padded outputs of the stacked
// decoder heads -> the (M, C) scene tensor)
struct ColSpans { dsc_col_span sp[DSC_SMALLK_MAX]; int count; int total; };
__global__ __launch_bounds__(256) void gather_columns_kernel(float* __restrict__ dst, int64_t ldd, const float* __restrict__ src, int64_t lds,
                                                             int rows, const ColSpans cs) {
    const long n = (long)rows * cs.total;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cs.total;
        int c = (int)(i - r * cs.total);
#pragma unroll
        for (int k = 0; k < DSC_SMALLK_MAX; ++k) {
            if (k < cs.count) {
                if (c >= 0 && c < cs.sp[k].width) dst[r * ldd + cs.sp[k].dst_col + c] = src[r * lds + cs.sp[k].src_col + c];
                c -= cs.sp[k].width;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
__global__ void time_embedding_kernel(const int64_t* __restrict__ t, int b, int dim, const float* __restrict__ table,
                                      int table_rows, const float* __restrict__ freq, float* __restrict__ out) {
    const int row = blockIdx.x;
    if (row >= b) return;
    const int64_t tv = t[row];
    const int half = dim >> 1;
    for (int c = threadIdx.x; c < dim; c += blockDim.x) {
        float o;
        if (table && tv >= 0 && tv < table_rows) o = table[tv * dim + c];
        else {
            const float a = (float)tv * freq[c < half ? c : c - half];
            o = c < half ? sinf(a) : cosf(a);
        }
        out[(int64_t)row * dim + c] = o;
    }
}

__global__ void activation_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t count, int act) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) y[i] = dsc_act(x[i], act);
}

}  // namespace

extern "C" int dsc_version(void) { return 100; }

extern "C" int dsc_weight_standardize_f32(const dsc_ws_item* items, int32_t count, float eps, dsc_stream_t stream) {
    if (!items || count < 1 || count > DSC_WS_MAX) return DSC_EINVAL;
    WsBatch b;
    int maxrows = 0;
    for (int i = 0; i < count; ++i) {
        if (!items[i].w || !items[i].out || items[i].rows < 1 || items[i].cols < 1) return DSC_EINVAL;
        if (items[i].cols > 2048) return DSC_ERANGE;
        b.it[i] = items[i];
        if (items[i].rows > maxrows) maxrows = items[i].rows;
    }
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(ws_kernel, dim3(maxrows, count), dim3(256), 0, static_cast<hipStream_t>(stream), b, eps);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_layernorm_f32(const float* x, int64_t ldx, const float* g, const float* residual, int64_t ldr,
                                 float* y, int64_t ldy, int32_t m, int32_t d, float eps, dsc_stream_t stream) {
    if (!x || !g || !y || m < 1) return DSC_EINVAL;
    if (d != 512) return DSC_ERANGE;
    if (!dsc_aligned16(x) || !dsc_aligned16(g) || !dsc_aligned16(y) || (ldx & 3) || (ldy & 3)) return DSC_EALIGN;
    if (residual && (!dsc_aligned16(residual) || (ldr & 3))) return DSC_EALIGN;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(layernorm512_kernel, dim3((m + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, ldx, g, residual, ldr, y, ldy, m, eps);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_linear_attention_f32(const float* q, int64_t ldq, const float* k, int64_t ldk,
                                        const float* v, int64_t ldv, float* out, int64_t ldo,
                                        int32_t scenes, int32_t nq, int32_t nk, float scale, dsc_stream_t stream) {
    if (!q || !k || !v || !out || scenes < 1 || nq < 1 || nk < 1) return DSC_EINVAL;
    if (nk > MAXTOK) return DSC_ERANGE;
    if (!dsc_aligned16(q) || !dsc_aligned16(k) || !dsc_aligned16(v) || !dsc_aligned16(out) ||
        (ldq & 3) || (ldk & 3) || (ldv & 3) || (ldo & 3)) return DSC_EALIGN;
    if (nq > MAXTOK) return DSC_ERANGE;
    const size_t lds = sizeof(float) * ((size_t)(2 * nk + nq) * LP + 32 * LP);      // <= 74 KB at 160 tokens
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_attention_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(linear_attention_kernel, dim3(scenes * DSC_HEADS), dim3(256), lds,
                       static_cast<hipStream_t>(stream), q, ldq, k, ldk, v, ldv, out, ldo, nq, nk, scale);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_attention_f32(const float* q, int64_t ldq, const float* k, int64_t ldk,
                                 const float* v, int64_t ldv, float* out, int64_t ldo,
                                 int32_t scenes, int32_t n, float scale, dsc_stream_t stream) {
    if (!q || !k || !v || !out || scenes < 1 || n < 1) return DSC_EINVAL;
    if (n > MAXTOK) return DSC_ERANGE;
    if (!dsc_aligned16(q) || !dsc_aligned16(k) || !dsc_aligned16(v) || !dsc_aligned16(out) ||
        (ldq & 3) || (ldk & 3) || (ldv & 3) || (ldo & 3)) return DSC_EALIGN;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(attention_kernel, dim3(scenes * DSC_HEADS), dim3(dsc_attention_threads(n)), (size_t)2 * n * LP * sizeof(float),
                       static_cast<hipStream_t>(stream), q, ldq, k, ldk, v, ldv, out, ldo, n, scale);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_linear_smallk_f32(const float* x, int64_t ldx, int32_t k_in, const float* w, int64_t ldw,
                                     const float* bias, float* y, int64_t ldy, int32_t m, int32_t n,
                                     int32_t act_out, dsc_stream_t stream) {
    if (!x || !w || !y || m < 1 || n < 1 || k_in < 1) return DSC_EINVAL;
    if (k_in > SK_MAXK) return DSC_ERANGE;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(linear_smallk_kernel, dim3((m + SK_TOK - 1) / SK_TOK), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, ldx, k_in, w, ldw, bias, y, ldy, m, n, act_out);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_linear_smallk_grouped_f32(const dsc_smallk_item* items, int32_t count, int32_t m, int32_t n, int32_t act_out,
                                             dsc_stream_t stream) {
    if (!items || count < 1 || count > DSC_SMALLK_MAX || m < 1 || n < 1) return DSC_EINVAL;
    SmallKBatch b;
    for (int i = 0; i < count; ++i) {
        if (!items[i].x || !items[i].w || !items[i].y || items[i].k_in < 1) return DSC_EINVAL;
        if (items[i].k_in > SK_MAXK) return DSC_ERANGE;
        b.it[i] = items[i];
    }
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(linear_smallk_grouped_kernel, dim3((m + SK_TOK - 1) / SK_TOK, count), dim3(256), 0,
                       static_cast<hipStream_t>(stream), b, m, n, act_out);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_gather_columns_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int32_t rows, const dsc_col_span* spans,
                                      int32_t count, dsc_stream_t stream) {
    if (!dst || !src || !spans || rows < 1 || count < 1 || count > DSC_SMALLK_MAX) return DSC_EINVAL;
    ColSpans cs;
    cs.count = count;
    cs.total = 0;
    for (int i = 0; i < count; ++i) {
        if (spans[i].width < 1 || spans[i].src_col < 0 || spans[i].dst_col < 0) return DSC_EINVAL;
        cs.sp[i] = spans[i];
        cs.total += spans[i].width;
    }
    long blocks = ((long)rows * cs.total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(gather_columns_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), dst, ldd, src, lds,
                       rows, cs);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_time_embedding_f32(const int64_t* t, int32_t b, int32_t dim, const float* table, int32_t table_rows,
                                      const float* freq, float* out, dsc_stream_t stream) {
    if (!t || !freq || !out || b < 1 || dim < 2 || (dim & 1)) return DSC_EINVAL;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(time_embedding_kernel, dim3(b), dim3(256), 0, static_cast<hipStream_t>(stream),
                       t, b, dim, table, table_rows, freq, out);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_activation_f32(const float* x, float* y, int64_t count, int32_t act, dsc_stream_t stream) {
    if (!x || !y || count < 1) return DSC_EINVAL;
    int64_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(activation_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, y, count, act);
    DSC_LAUNCH_CHECK();
    return 0;
}

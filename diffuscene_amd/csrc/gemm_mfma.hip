// fp32 MFMA GEMM with fused epilogues for the Unet1D denoiser (gfx950).
//
// Every 1x1 conv / linear of the reference denoiser (denoise_net.py) is  Y[m][n] = X[m][k] . W[n][k]^T:
// both operands K-contiguous.  The kernel computes the TRANSPOSED product with
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain):  D[channel][token] = sum_k W[channel][k] X[token][k]
//   A operand (32 "rows")  = 32 output channels of W,   lane l supplies W[c0 + (l&31)][k + (l>>5)]
//   B operand (32 "cols")  = 32 tokens of X,            lane l supplies X[t0 + (l&31)][k + (l>>5)]
//   D: lane l holds token t0 + (l&31) and channels c0 + (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15.
// In that orientation every per-token reduction over channels (LayerNorm-like) is in-lane, a lane's
// register quad is 4 CONSECUTIVE channels (16-byte stores), and all tokens of a scene sit in lanes of the
// same block, so GroupNorm (64 channels x N tokens of one scene, denoise_net.py:164) is computed from the
// accumulators: the whole Block.forward (WS-conv + GroupNorm + scale/shift + SiLU, :167-176) is one kernel.
//
// Tiling: 256 threads = 4 waves; block tile BM tokens x 128 channels; BK = 32 staged through LDS with a
// padded row stride of 36 floats (ds_read_b128 of 16 rows is bank-conflict-free: 36*r mod 64 hits 16
// distinct 4-bank slots).  A lane reads 4 consecutive k (one ds_read_b128) per fragment and feeds 4 MFMAs
// (lanes 0-31 carry k..k+3, lanes 32-63 carry k+4..k+7: the k-permutation is the same for both operands).
// Register-staged prefetch of tile kt+1 overlaps the MFMAs of tile kt; two blocks per CU cover barriers.
#include "dsc_common.h"

namespace {

constexpr int BK = 32;
constexpr int LDT = 36;

template <int TM, int TN, int WM, int WN, bool GN>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const dsc_gemm_args p, const int ncolblk) {
    constexpr int BM = 32 * TM * WM;
    constexpr int BN = 32 * TN * WN;
    static_assert(WM * WN == 4, "4 waves per block");
    constexpr int XF = BM * 8 / 256;
    constexpr int WF = BN * 8 / 256;
    static_assert(XF * 256 == BM * 8 && WF * 256 == BN * 8, "tile must split evenly over 256 threads");

    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDT];
    float* Xs = smem;
    float* Ws = smem + BM * LDT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;

    const int z = blockIdx.y;
    const int rb = blockIdx.x / ncolblk, cb = blockIdx.x % ncolblk;

    const int N = GN ? p.tokens_per_scene : 1;
    const int rows_per_blk = GN ? (BM / N) * N : BM;
    const int64_t row0 = (int64_t)rb * rows_per_blk;
    const int rows_here = (int)((p.m - row0) < (int64_t)rows_per_blk ? (p.m - row0) : (int64_t)rows_per_blk);
    const int col0 = cb * BN;

    const float* a1 = p.a1 + (int64_t)z * p.sa1;
    const float* a2 = p.a2 ? p.a2 + (int64_t)z * p.sa2 : nullptr;
    const float* w = p.w + (int64_t)z * p.sw;
    const float* bias = p.bias ? p.bias + (int64_t)z * p.sbias : nullptr;
    const float* res = p.residual ? p.residual + (int64_t)z * p.sres : nullptr;
    float* y = p.y + (int64_t)z * p.sy;

    const int nk = (p.k1 + p.k2) / BK;

    f32x4 xr[XF], wr[WF];
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.0f;

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        const float* ab;
        int64_t lda;
        int kk;
        if (k0 < p.k1) { ab = a1; lda = p.lda1; kk = k0; }
        else           { ab = a2; lda = p.lda2; kk = k0 - p.k1; }
#pragma unroll
        for (int i = 0; i < XF; ++i) {
            const int f = tid + 256 * i;
            const int r = f >> 3, kq = f & 7;
            if (r < rows_here) xr[i] = *reinterpret_cast<const f32x4*>(ab + (row0 + r) * lda + kk + kq * 4);
            else               xr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < WF; ++i) {
            const int f = tid + 256 * i;
            const int r = f >> 3, kq = f & 7;
            const int c = col0 + r;
            if (c < p.n) wr[i] = *reinterpret_cast<const f32x4*>(w + (int64_t)c * p.ldw + k0 + kq * 4);
            else         wr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < XF; ++i) {
            const int f = tid + 256 * i;
            const int r = f >> 3, kq = f & 7;
            f32x4 v = xr[i];
            if (p.act_in == DSC_ACT_SILU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = dsc_silu(v[e]);
            }
            *reinterpret_cast<f32x4*>(Xs + r * LDT + kq * 4) = v;
        }
#pragma unroll
        for (int i = 0; i < WF; ++i) {
            const int f = tid + 256 * i;
            const int r = f >> 3, kq = f & 7;
            *reinterpret_cast<f32x4*>(Ws + r * LDT + kq * 4) = wr[i];
        }
    };

    load_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
        store_tile();
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            f32x4 xf[TM], wf[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                xf[tm] = *reinterpret_cast<const f32x4*>(Xs + ((wm * TM + tm) * 32 + l31) * LDT + k8 * 8 + half * 4);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                wf[tn] = *reinterpret_cast<const f32x4*>(Ws + ((wn * TN + tn) * 32 + l31) * LDT + k8 * 8 + half * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[tn][j], xf[tm][j], acc[tm][tn], 0, 0, 0);
        }
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue
    // bias
    if (bias) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = col0 + (wn * TN + tn) * 32 + 8 * q + 4 * half;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float b = (c + e < p.n) ? bias[c + e] : 0.0f;
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) acc[tm][tn][4 * q + e] += b;
                }
            }
    }

    const bool vec_y = ((p.ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0) && ((p.n & 3) == 0);
    const bool vec_r = res && ((p.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(res) & 15) == 0) && ((p.n & 3) == 0);

    if constexpr (GN) {
        if (p.preact) {                   // training: keep the pre-norm conv output for the backward pass
            float* zp = p.preact + (int64_t)z * p.sy;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int tl = (wm * TM + tm) * 32 + l31;
                if (tl >= rows_here) continue;
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = col0 + (wn * TN + tn) * 32 + 8 * q + 4 * half;
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[tm][tn][4 * q + e];
                        *reinterpret_cast<f32x4*>(zp + (row0 + tl) * p.ld_preact + c) = v;
                    }
            }
        }
        constexpr int G = BN / 64;        // GroupNorm groups covered by this block
        constexpr int CT = BN / 32;       // 32-channel tiles in the block
        float* P = smem;                  // [CT][BM] per-token partial sums
        float* stat = smem + CT * BM;     // [spt*G] mean, then [spt*G] rstd
        const int spt = BM / N;
        const int scenes_here = rows_here / N;
        const int nstat = scenes_here * G;
        const float inv_cnt = 1.0f / (64.0f * (float)N);
        int scn[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int tl = (wm * TM + tm) * 32 + l31;
            scn[tm] = tl / N;
        }
        // pass 1: mean
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int tl = (wm * TM + tm) * 32 + l31;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[tm][tn][r];
                s += __shfl_xor(s, 32, 64);
                if (half == 0) P[(wn * TN + tn) * BM + tl] = s;
            }
        }
        __syncthreads();
        for (int st = wave; st < nstat; st += 4) {
            const int sc = st / G, g = st % G;
            float s = 0.f;
            for (int j = lane; j < 2 * N; j += 64) {
                const int ct = 2 * g + (j >= N ? 1 : 0);
                const int tk = (j >= N ? j - N : j);
                s += P[ct * BM + sc * N + tk];
            }
            s = wave_sum(s);
            if (lane == 0) stat[st] = s * inv_cnt;
        }
        __syncthreads();
        // pass 2: variance about the mean
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int tl = (wm * TM + tm) * 32 + l31;
            const bool ok = scn[tm] < scenes_here;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int g = (wn * TN + tn) >> 1;
                const float mu = ok ? stat[scn[tm] * G + g] : 0.f;
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float d = acc[tm][tn][r] - mu; s += d * d; }
                s += __shfl_xor(s, 32, 64);
                if (half == 0) P[(wn * TN + tn) * BM + tl] = s;
            }
        }
        __syncthreads();
        for (int st = wave; st < nstat; st += 4) {
            const int sc = st / G, g = st % G;
            float s = 0.f;
            for (int j = lane; j < 2 * N; j += 64) {
                const int ct = 2 * g + (j >= N ? 1 : 0);
                const int tk = (j >= N ? j - N : j);
                s += P[ct * BM + sc * N + tk];
            }
            s = wave_sum(s);
            if (lane == 0) stat[spt * G + st] = 1.0f / sqrtf(s * inv_cnt + p.eps);
        }
        __syncthreads();
        // normalise, affine, scale/shift, SiLU, residual, store
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int tl = (wm * TM + tm) * 32 + l31;
            if (tl >= rows_here) continue;
            const int64_t tok = row0 + tl;
            int64_t ssrow = 0;
            if (p.ss_mode == DSC_SS_PER_TOKEN) ssrow = tok;
            else if (p.ss_mode == DSC_SS_PER_SCENE) ssrow = tok / N;
            else if (p.ss_mode == DSC_SS_PER_SLOT) ssrow = tok % N;
            const float* ss = p.scale_shift ? p.scale_shift + ssrow * p.ld_ss : nullptr;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int g = (wn * TN + tn) >> 1;
                const float mu = stat[scn[tm] * G + g];
                const float rs = stat[spt * G + scn[tm] * G + g];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = col0 + (wn * TN + tn) * 32 + 8 * q + 4 * half;
                    const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
                    const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (acc[tm][tn][4 * q + e] - mu) * rs * ga[e] + be[e];
                    if (ss) {
                        const f32x4 sc4 = *reinterpret_cast<const f32x4*>(ss + c);
                        const f32x4 sh4 = *reinterpret_cast<const f32x4*>(ss + p.n + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] * (sc4[e] + 1.0f) + sh4[e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = dsc_silu(v[e]);
                    if (res) {
                        if (vec_r) {
                            const f32x4 r4 = *reinterpret_cast<const f32x4*>(res + tok * p.ldr + c);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += r4[e];
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += res[tok * p.ldr + c + e];
                        }
                    }
                    if (vec_y) *reinterpret_cast<f32x4*>(y + tok * p.ldy + c) = v;
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[tok * p.ldy + c + e] = v[e];
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int tl = (wm * TM + tm) * 32 + l31;
            if (tl >= rows_here) continue;
            const int64_t tok = row0 + tl;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = col0 + (wn * TN + tn) * 32 + 8 * q + 4 * half;
                    if (c >= p.n) continue;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = dsc_act(acc[tm][tn][4 * q + e], p.act_out);
                    if (c + 3 < p.n) {
                        if (res) {
                            if (vec_r) {
                                const f32x4 r4 = *reinterpret_cast<const f32x4*>(res + tok * p.ldr + c);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += r4[e];
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += res[tok * p.ldr + c + e];
                            }
                        }
                        if (vec_y) *reinterpret_cast<f32x4*>(y + tok * p.ldy + c) = v;
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) y[tok * p.ldy + c + e] = v[e];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (c + e < p.n) {
                                float o = v[e];
                                if (res) o += res[tok * p.ldr + c + e];
                                y[tok * p.ldy + c + e] = o;
                            }
                    }
                }
            }
        }
    }
}

int check_common(const dsc_gemm_args* a) {
    if (!a || !a->a1 || !a->w || !a->y) return DSC_EINVAL;
    if (a->m <= 0 || a->n <= 0 || a->k1 <= 0 || a->k2 < 0) return DSC_EINVAL;
    if ((a->k1 % BK) || (a->k2 % BK)) return DSC_EINVAL;
    if (a->k2 > 0 && !a->a2) return DSC_EINVAL;
    if (a->batch < 1) return DSC_EINVAL;
    if (!dsc_aligned16(a->a1) || (a->lda1 & 3) || !dsc_aligned16(a->w) || (a->ldw & 3)) return DSC_EALIGN;
    if (a->k2 > 0 && (!dsc_aligned16(a->a2) || (a->lda2 & 3))) return DSC_EALIGN;
    if (a->batch > 1 && ((a->sa1 & 3) || (a->sa2 & 3) || (a->sw & 3))) return DSC_EALIGN;
    if (a->act_in != DSC_ACT_NONE && a->act_in != DSC_ACT_SILU) return DSC_EINVAL;
    return 0;
}

template <int TM, int TN, int WM, int WN, bool GN>
int launch(const dsc_gemm_args* a, int rows_per_blk, hipStream_t s) {
    constexpr int BN = 32 * TN * WN;
    const int nrb = (a->m + rows_per_blk - 1) / rows_per_blk;
    const int ncb = (a->n + BN - 1) / BN;
    dim3 grid((unsigned)(nrb * ncb), (unsigned)a->batch);
    hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, GN>), grid, dim3(256), 0, s, *a, ncb);
    DSC_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int dsc_gemm_f32(const dsc_gemm_args* a, dsc_stream_t stream) {
    int rc = check_common(a);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // pick the token-tile height that wastes the fewest padded rows (ties -> taller tile)
    const int cand[3] = {160, 128, 96};
    int best = 160;
    long best_cost = -1;
    for (int i = 0; i < 3; ++i) {
        const long cost = (long)((a->m + cand[i] - 1) / cand[i]) * cand[i];
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = cand[i]; }
    }
    if (best == 160) return launch<5, 1, 1, 4, false>(a, 160, s);
    if (best == 128) return launch<2, 2, 2, 2, false>(a, 128, s);
    return launch<3, 1, 1, 4, false>(a, 96, s);
}

extern "C" int dsc_gemm_gn_silu_f32(const dsc_gemm_args* a, dsc_stream_t stream) {
    int rc = check_common(a);
    if (rc) return rc;
    if (!a->gamma || !a->beta || !a->bias) return DSC_EINVAL;
    const int N = a->tokens_per_scene;
    if (N < 1 || N > 160) return DSC_ERANGE;
    if (a->m % N) return DSC_EINVAL;
    if (a->n % 128) return DSC_EINVAL;
    if (a->act_in != DSC_ACT_NONE) return DSC_EINVAL;
    if (!dsc_aligned16(a->gamma) || !dsc_aligned16(a->beta)) return DSC_EALIGN;
    if (a->preact && (!dsc_aligned16(a->preact) || (a->ld_preact & 3))) return DSC_EALIGN;
    if (a->ss_mode != DSC_SS_NONE) {
        if (!a->scale_shift) return DSC_EINVAL;
        if (!dsc_aligned16(a->scale_shift) || (a->ld_ss & 3)) return DSC_EALIGN;
        if (a->ss_mode < DSC_SS_NONE || a->ss_mode > DSC_SS_PER_SLOT) return DSC_EINVAL;
    } else if (a->scale_shift) return DSC_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int cand[3] = {160, 128, 96};
    int best = 0;
    double best_u = -1.0;
    for (int i = 0; i < 3; ++i) {
        if (cand[i] < N) continue;
        const double u = (double)((cand[i] / N) * N) / cand[i];
        if (u > best_u + 1e-9) { best_u = u; best = cand[i]; }
    }
    if (!best) return DSC_ERANGE;
    const int rpb = (best / N) * N;
    if (best == 160) return launch<5, 1, 1, 4, true>(a, rpb, s);
    if (best == 128) return launch<2, 2, 2, 2, true>(a, rpb, s);
    return launch<3, 1, 1, 4, true>(a, rpb, s);
}

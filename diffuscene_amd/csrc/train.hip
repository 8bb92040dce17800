// Backward kernels of the DDPM training step (reference: autograd through denoise_net.py; here hand-written).
//
//   gemm_tn_kernel        dW[n][k] = sum_m dY[m][n] * A[m][k]   (weight gradients; fp32 MFMA, split over tokens)
//   reduce_slabs_kernel   deterministic second stage of every split reduction
//   colsum_kernel         bias / affine gradients: column sums over tokens
//   gn_silu_bwd_kernel    backward of the fused GroupNorm + (scale+1,shift) + SiLU epilogue, one (scene, group) per block
//   ws_bwd_kernel         backward of weight standardisation
//   layernorm_bwd_kernel  backward of the channel LayerNorm (+ gain gradient partials)
//   linear_attention_bwd_kernel / attention_bwd_kernel   one (scene, head) per block
//   act_bwd_kernel, transpose_kernel
// Input gradients (dA = dY . W) reuse the forward NT kernel (gemm_mfma.hip) on transposed weights.
#define DSC_BAD_INDEX_COUNTER
#include "dsc_common.h"

namespace {

// =====================================================================================================
// TN GEMM: out[j = n][i = k] = sum_m dy[m][n] * a[m][k].   MFMA A operand = 32 input channels (i), B operand = 32
// output channels (j), reduction over tokens m two at a time.  Both tiles are staged exactly as they lie in HBM
// ([m][channel], channel contiguous), fragments are ds_read_b32 along the channel -> conflict-free, no transposes.
// =====================================================================================================
constexpr int TN_BK = 32;      // tokens per staged tile
constexpr int TN_LD = 132;     // 128 + 4 pad floats

struct TnArgs {
    const float* a1; long lda1; int k1;
    const float* a2; long lda2; int k2;       // optional second channel segment
    const float* dy; long ldd;
    float* out; long ldo;                     // [n][k1+k2] (or slab s at out + s*slab)
    float* bias_out;                          // optional: column sums of dy (bias gradient), [n] (or slab s)
    long slab;                                // elements between split slabs (0 when splits == 1)
    long bias_slab;                           // elements between bias slabs
    int m, n, kvalid;                         // kvalid: columns >= kvalid are not stored (padded small-K inputs)
    int chunk;                                // tokens per split (multiple of 32)
    int ktiles;                               // number of 128-wide k tiles
};

__device__ __forceinline__ void tn_block(const TnArgs& p, const int it, const int jt, const int split, float* As, float* Bs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;
    const int i0 = it * 128, j0 = jt * 128;
    const long m_begin = (long)split * p.chunk;
    const long m_end = (m_begin + p.chunk < p.m) ? m_begin + p.chunk : p.m;

    const float* ab;
    long lda;
    int ic;          // column of this k tile inside its segment
    int kseg;        // columns available in the segment
    if (i0 < p.k1) { ab = p.a1; lda = p.lda1; ic = i0; kseg = p.k1; }
    else           { ab = p.a2; lda = p.lda2; ic = i0 - p.k1; kseg = p.k2; }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const bool do_bias = p.bias_out && it == 0 && tid < 128;   // the first k-tile column of blocks also owns the bias gradient
    float bsum = 0.f;
    f32x4 ar[4], br[4];
    auto load_tile = [&](long m0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f = tid + 256 * q;
            const int r = f >> 5, c4 = (f & 31) * 4;
            // branch-free: out-of-range elements read a valid address and are zeroed when staged one tile later
            const long m = (m0 + r < m_end) ? m0 + r : m_begin;
            ar[q] = *reinterpret_cast<const f32x4*>(ab + m * lda + ic + ((ic + c4 < kseg) ? c4 : 0));
            br[q] = *reinterpret_cast<const f32x4*>(p.dy + m * p.ldd + j0 + ((j0 + c4 < p.n) ? c4 : 0));
        }
    };
    auto store_tile = [&](long m0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f = tid + 256 * q;
            const int r = f >> 5, c4 = (f & 31) * 4;
            const bool mok = m0 + r < m_end;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(As + r * TN_LD + c4) = (mok && ic + c4 < kseg) ? ar[q] : z;
            *reinterpret_cast<f32x4*>(Bs + r * TN_LD + c4) = (mok && j0 + c4 < p.n) ? br[q] : z;
        }
    };

    if (m_begin < m_end) {
        load_tile(m_begin);
        for (long m0 = m_begin; m0 < m_end; m0 += TN_BK) {
            store_tile(m0);
            __syncthreads();
            if (m0 + TN_BK < m_end) load_tile(m0 + TN_BK);
            if (do_bias) {
#pragma unroll
                for (int r = 0; r < TN_BK; ++r) bsum += Bs[r * TN_LD + tid];
            }
#pragma unroll
            for (int s = 0; s < TN_BK / 2; ++s) {
                float af[2], bf[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) af[a] = As[(2 * s + half) * TN_LD + (wi * 2 + a) * 32 + l31];
#pragma unroll
                for (int b = 0; b < 2; ++b) bf[b] = Bs[(2 * s + half) * TN_LD + (wj * 2 + b) * 32 + l31];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
            __syncthreads();
        }
    }
    float* out = p.out + (long)split * p.slab;
    const bool vec = (p.ldo & 3) == 0 && (p.slab & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;
    if (do_bias && j0 + tid < p.n) p.bias_out[(long)split * p.bias_slab + j0 + tid] = bsum;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int j = j0 + (wj * 2 + b) * 32 + l31;          // output row (n)
        if (j >= p.n) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + (wi * 2 + a) * 32 + 8 * q + 4 * half;   // output column (k), 4 consecutive
                if (vec && i + 3 < p.kvalid) {
                    f32x4 t4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) t4[e] = acc[a][b][4 * q + e];
                    *reinterpret_cast<f32x4*>(out + (long)j * p.ldo + i) = t4;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (i + e < p.kvalid) out[(long)j * p.ldo + i + e] = acc[a][b][4 * q + e];
                }
            }
    }
}


// -----------------------------------------------------------------------------------------------------
// The same tile with the interleaved LDS-DMA main loop of gemm_core.h (IL == 2): full 32-token tiles go from HBM straight into
// one of two LDS stages (buffer_load_dwordx4 ... lds, 1 KiB = 2 token rows x 512 B per wave instruction), the fragment reads of
// step s+1 and the DMA of tile kt+1 are slotted between the MFMAs of step s, one block barrier per tile.  LDS rows are unpadded
// (128 floats); the 32-float channel group g of token row r sits at position g ^ (r & 1), so the two token rows a fragment
// read touches (lanes 0-31: row 2s, lanes 32-63: row 2s+1) fall into different bank halves.  A ragged last tile (tokens that
// must contribute zeros) is staged through registers with masking, as the classic loop does.
// -----------------------------------------------------------------------------------------------------
constexpr int TN_OPF = TN_BK * 128;            // floats per operand per stage
constexpr int TN_STAGE = 2 * TN_OPF;           // A rows | dY rows

__device__ __forceinline__ void tn_block_dma(const TnArgs& p, const int it, const int jt, const int split, float* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;
    const int i0 = it * 128, j0 = jt * 128;
    const long m_begin = (long)split * p.chunk;
    const long m_end = (m_begin + p.chunk < p.m) ? m_begin + p.chunk : p.m;

    const float* ab;
    long lda;
    int ic, kseg;
    if (i0 < p.k1) { ab = p.a1; lda = p.lda1; ic = i0; kseg = p.k1; }
    else           { ab = p.a2; lda = p.lda2; ic = i0 - p.k1; kseg = p.k2; }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const bool do_bias = p.bias_out && it == 0 && tid < 128;   // the first k-tile column of blocks also owns the bias gradient
    float bsum = 0.f;
    // fragment addresses (floats, relative to an operand's stage): row 2s + half, channel group (w*2 + t) at position ^ half
    int fa[2], fb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        fa[t] = half * 128 + ((((wi * 2 + t) ^ half) << 5) | l31);
        fb[t] = TN_OPF + half * 128 + ((((wj * 2 + t) ^ half) << 5) | l31);
    }
    const int bcol = ((tid >> 5) << 5), bl = tid & 31;          // bias column sums: column tid of the dY tile

    // DMA chunk c of a wave = token rows 2c, 2c+1: lanes 0-31 row 2c, lanes 32-63 row 2c+1; slot (lane & 31) of the LDS row holds
    // the quad of channel group ((slot >> 3) ^ (row & 1))
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int va[4], vb[4];
    {
        const int slot = lane & 31, rpar = lane >> 5;
        const int col = ((((slot >> 3) ^ rpar) << 3) | (slot & 7)) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 2 * (wave_u + 4 * i) + rpar;
            va[i] = (row * (int)lda + col) * 4;
            vb[i] = (row * (int)p.ldd + col) * 4;
        }
    }
    const long full_end = m_begin + ((m_end - m_begin) / TN_BK) * TN_BK;    // tokens covered by full tiles
    auto dma_tile = [&](long m0, float* stage) {
#if defined(__HIP_DEVICE_COMPILE__)
        // descriptors rebuilt per tile (scalar work): base = first token row of the tile, num_records = bytes up to the end of
        // the matrix, so channel quads past a ragged K / n read the following rows (never stored) or, at the very end, zeros
        const float* abase = ab + m0 * lda + ic;
        const float* dbase = p.dy + m0 * p.ldd + j0;
        long arec = ((p.m - 1 - m0) * lda + (kseg - ic)) * 4, drec = ((p.m - 1 - m0) * p.ldd + (p.n - j0)) * 4;
        if (arec > 0x7fffffffL) arec = 0x7fffffffL;      // the bound only matters within 64 rows of the end of the matrix
        if (drec > 0x7fffffffL) drec = 0x7fffffffL;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(abase), 0, (int)arec, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dbase), 0, (int)drec, 0x00020000);
        __attribute__((address_space(3))) char* lb = (__attribute__((address_space(3))) char*)stage;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, lb + (wave_u + 4 * i) * 1024, 16, va[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, lb + TN_OPF * 4 + (wave_u + 4 * i) * 1024, 16, vb[i], 0, 0, 0);
#else
        (void)m0; (void)stage;
#endif
    };
    auto bias_rows = [&](const float* stage) {
        if (do_bias) {
#pragma unroll
            for (int r = 0; r < TN_BK; ++r) bsum += stage[TN_OPF + r * 128 + ((bcol ^ ((r & 1) << 5)) | bl)];
        }
    };

    if (m_begin < full_end) {
        float afA[2], bfA[2], afB[2], bfB[2];
        auto frags = [&](const float* stage, int s_, float (&af)[2], float (&bf)[2]) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = stage[fa[t] + s_ * 256];
                bf[t] = stage[fb[t] + s_ * 256];
            }
        };
        auto mma = [&](const float (&af)[2], const float (&bf)[2]) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        };
        dma_tile(m_begin, smem);
        __builtin_amdgcn_s_waitcnt(0x0f70);       // vmcnt(0)
        __syncthreads();
        frags(smem, 0, afA, bfA);
        int kt = 0;
        for (long m0 = m_begin; m0 < full_end; m0 += TN_BK, ++kt) {
            const float* cur = smem + (kt & 1) * TN_STAGE;
            float* nxt = smem + ((kt + 1) & 1) * TN_STAGE;
            const long mn = (m0 + TN_BK < full_end) ? m0 + TN_BK : m0;     // the last iteration re-fetches its own tile (unused)
            bias_rows(cur);
            __builtin_amdgcn_sched_barrier(0);
            // steps 0..7 carry the DMA of the next tile (8 instructions per wave), every step the fragment reads of the next one
            dma_tile(mn, nxt);
#pragma unroll
            for (int s2 = 0; s2 < TN_BK / 4 - 1; ++s2) {
                frags(cur, 2 * s2 + 1, afB, bfB);
                mma(afA, bfA);
                frags(cur, 2 * s2 + 2, afA, bfA);
                mma(afB, bfB);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int i = 0; i < (TN_BK / 4 - 1) * 8 - 32; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // last two steps: 14 (A) and 15 (B); the barrier sits between them
            frags(cur, TN_BK / 2 - 1, afB, bfB);
            mma(afA, bfA);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0x0f70);   // this wave's DMA of the next tile has landed
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            frags(nxt, 0, afA, bfA);
            mma(afB, bfB);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    if (full_end < m_end) {
        // ragged tail (< 32 tokens): register-staged with zero masking into stage 0, plain compute
        float* As = smem;
        float* Bs = smem + TN_OPF;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f = tid + 256 * q;
            const int r = f >> 5, c4 = (f & 31) * 4;
            const bool mok = full_end + r < m_end;
            const long m = mok ? full_end + r : m_begin;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 av = *reinterpret_cast<const f32x4*>(ab + m * lda + ic + ((ic + c4 < kseg) ? c4 : 0));
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.dy + m * p.ldd + j0 + ((j0 + c4 < p.n) ? c4 : 0));
            const int pos = r * 128 + ((((c4 >> 5) ^ (r & 1)) << 5) | (c4 & 31));
            *reinterpret_cast<f32x4*>(As + pos) = (mok && ic + c4 < kseg) ? av : z;
            *reinterpret_cast<f32x4*>(Bs + pos) = (mok && j0 + c4 < p.n) ? bv : z;
        }
        __syncthreads();
        bias_rows(smem);
#pragma unroll
        for (int s_ = 0; s_ < TN_BK / 2; ++s_) {
            float af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = smem[fa[t] + s_ * 256];
                bf[t] = smem[fb[t] + s_ * 256];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
    }
    float* out = p.out + (long)split * p.slab;
    const bool vec = (p.ldo & 3) == 0 && (p.slab & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;
    if (do_bias && j0 + tid < p.n) p.bias_out[(long)split * p.bias_slab + j0 + tid] = bsum;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int j = j0 + (wj * 2 + b) * 32 + l31;          // output row (n)
        if (j >= p.n) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + (wi * 2 + a) * 32 + 8 * q + 4 * half;   // output column (k), 4 consecutive
                if (vec && i + 3 < p.kvalid) {
                    f32x4 t4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) t4[e] = acc[a][b][4 * q + e];
                    *reinterpret_cast<f32x4*>(out + (long)j * p.ldo + i) = t4;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (i + e < p.kvalid) out[(long)j * p.ldo + i + e] = acc[a][b][4 * q + e];
                }
            }
    }
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const TnArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * TN_STAGE];
    tn_block_dma(p, blockIdx.x % p.ktiles, blockIdx.x / p.ktiles, blockIdx.y, smem);
}

// -----------------------------------------------------------------------------------------------------
// Grouped TN GEMM: the weight gradients of MANY layers in one launch.  Weight gradients are leaves of the backward pass, and
// the static training plan keeps every activation alive, so they can all be deferred: with ~2200 output tiles in flight
// the token dimension needs (almost) no splitting -- the per-layer launch above has to cut M = 20480 into 32 slabs to fill
// 256 CUs with the 16 tiles of one 512 x 512 gradient, and then writes + re-reads 32 partial copies of every gradient.
// Tiles of one group are consecutive ids: they run concurrently, walk the tokens in step and share the A / dY rows through
// L2 / Infinity Cache.  The caller orders the groups longest first so that the short ones fill the tail.
// -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ int tn_find_group(const dsc_tn_group* __restrict__ g, int count, int tile) {
    int lo = 0, hi = count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (g[mid].tile0 <= tile) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256, 2) void gemm_tn_grouped_kernel(const dsc_tn_group* __restrict__ groups, const int count,
                                                                 const int splits, float* __restrict__ workspace) {
    __shared__ __attribute__((aligned(16))) float smem[2 * TN_STAGE];
    // no XCD remap: groups differ wildly in length (20480 tokens vs 256), so contiguous chunks per XCD would leave some XCDs
    // idle; round-robin dispatch spreads every group over all XCDs (tiles t and t+8 of a 512 x 512 gradient share their A rows
    // in one L2, the rest of the sharing happens in the Infinity Cache)
    const int tile = blockIdx.x;
    const int gi = tn_find_group(groups, count, tile);
    const dsc_tn_group g = groups[gi];
    const int K = g.k1 + g.k2;
    const int ktiles = (K + 127) / 128;
    const int local = tile - g.tile0;
    const int split = blockIdx.y;
    TnArgs p;
    p.a1 = g.a1; p.lda1 = g.lda1; p.k1 = g.k1; p.a2 = g.a2; p.lda2 = g.lda2; p.k2 = g.k2; p.dy = g.dy; p.ldd = g.ldd;
    p.m = g.m; p.n = g.n; p.kvalid = g.kvalid; p.ktiles = ktiles;
    p.chunk = ((g.m + splits - 1) / splits + 31) / 32 * 32;
    if (splits == 1) {
        p.out = g.out; p.ldo = g.ldo; p.bias_out = g.dbias; p.slab = 0; p.bias_slab = 0;
    } else {
        const long wslab = (long)g.n * g.kvalid;
        p.out = workspace + g.ws_offset; p.ldo = g.kvalid; p.slab = wslab;
        p.bias_out = g.dbias ? workspace + g.ws_offset + wslab * splits : nullptr; p.bias_slab = g.n;
    }
    tn_block_dma(p, local % ktiles, local / ktiles, split, smem);
}

// second stage of the grouped launch when splits > 1: every block sums the slabs of its own 128 x 128 tile (fixed order)
__global__ __launch_bounds__(256) void reduce_grouped_kernel(const dsc_tn_group* __restrict__ groups, const int count,
                                                             const int splits, const float* __restrict__ workspace) {
    const int tile = blockIdx.x;
    const int gi = tn_find_group(groups, count, tile);
    const dsc_tn_group g = groups[gi];
    const int K = g.k1 + g.k2;
    const int ktiles = (K + 127) / 128;
    const int local = tile - g.tile0;
    const int it = local % ktiles, jt = local / ktiles;
    const long wslab = (long)g.n * g.kvalid;
    const float* ws = workspace + g.ws_offset;
    const int i_lo = it * 128, i_hi = (i_lo + 128 < g.kvalid) ? i_lo + 128 : g.kvalid;
    const int j_lo = jt * 128, j_hi = (j_lo + 128 < g.n) ? j_lo + 128 : g.n;
    const int w = i_hi - i_lo;
    const bool vec = (w & 3) == 0 && (g.kvalid & 3) == 0 && (g.ldo & 3) == 0 && (g.ws_offset & 3) == 0 && (wslab & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(g.out) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0;
    if (w > 0 && vec) {
        // 32 lanes x 16 bytes cover a 128-wide row; 8 rows per pass
        const int c4 = (threadIdx.x & 31) * 4, r8 = threadIdx.x >> 5;
        if (c4 < w) {
            for (int j = j_lo + r8; j < j_hi; j += 8) {
                const float* src = ws + (long)j * g.kvalid + i_lo + c4;
                f32x4 acc = *reinterpret_cast<const f32x4*>(src);
                for (int s = 1; s < splits; ++s) acc += *reinterpret_cast<const f32x4*>(src + (long)s * wslab);
                *reinterpret_cast<f32x4*>(g.out + (long)j * g.ldo + i_lo + c4) = acc;
            }
        }
    } else if (w > 0) {
        for (int e = threadIdx.x; e < (j_hi - j_lo) * w; e += 256) {
            const int j = j_lo + e / w, i = i_lo + e % w;
            float acc = 0.f;
            for (int s = 0; s < splits; ++s) acc += ws[(long)s * wslab + (long)j * g.kvalid + i];
            g.out[(long)j * g.ldo + i] = acc;
        }
    }
    if (g.dbias && it == 0) {
        for (int j = j_lo + threadIdx.x; j < j_hi; j += 256) {
            float acc = 0.f;
            for (int s = 0; s < splits; ++s) acc += ws[wslab * splits + (long)s * g.n + j];
            g.dbias[j] = acc;
        }
    }
}

// out[i] = sum_s slabs[s][i]   (deterministic order)
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, long slab, int nslab, float* __restrict__ out,
                                    long count) {
    const long stride = (long)gridDim.x * blockDim.x;
    const long tid0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (((slab | count) & 3) == 0 && ((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        // 16-byte lanes, 4 slabs in flight per thread; slab order of the sum is fixed (k ascending within each of 4 chains)
        for (long i = tid0 * 4; i < count; i += stride * 4) {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
            int k = 0;
            for (; k + 3 < nslab; k += 4) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(slabs + (long)k * slab + i);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(slabs + (long)(k + 1) * slab + i);
                const f32x4 v2 = *reinterpret_cast<const f32x4*>(slabs + (long)(k + 2) * slab + i);
                const f32x4 v3 = *reinterpret_cast<const f32x4*>(slabs + (long)(k + 3) * slab + i);
                a0 += v0; a1 += v1; a2 += v2; a3 += v3;
            }
            for (; k < nslab; ++k) a0 += *reinterpret_cast<const f32x4*>(slabs + (long)k * slab + i);
            *reinterpret_cast<f32x4*>(out + i) = (a0 + a1) + (a2 + a3);
        }
        return;
    }
    for (long i = tid0; i < count; i += stride) {
        float s = 0.f;
        for (int k = 0; k < nslab; ++k) s += slabs[(long)k * slab + i];
        out[i] = s;
    }
}

// partial[s][c] = sum over rows of chunk s of x[r][c];  block = 64 columns x one chunk
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long ldx, int m, int n, int chunk,
                                                    float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rq = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.y * chunk;
    const long r1 = (r0 + chunk < m) ? r0 + chunk : m;
    float s = 0.f;
    if (c < n) {
        // 8 independent loads in flight per thread (the plain loop serialises on the load latency)
        const float* xp = x + (r0 + rq) * ldx + c;
        const long step = 4 * ldx;
        long r = r0 + rq;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (; r + 28 < r1; r += 32, xp += 8 * step) {
            const float v0 = xp[0], v1 = xp[step], v2 = xp[2 * step], v3 = xp[3 * step];
            const float v4 = xp[4 * step], v5 = xp[5 * step], v6 = xp[6 * step], v7 = xp[7 * step];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
            s0 += v4; s1 += v5; s2 += v6; s3 += v7;
        }
        for (; r < r1; r += 4, xp += step) s0 += xp[0];
        s = (s0 + s1) + (s2 + s3);
    }
    red[rq][threadIdx.x & 63] = s;
    __syncthreads();
    if (rq == 0 && c < n) partial[(long)blockIdx.y * n + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) +
                                                              (red[2][threadIdx.x] + red[3][threadIdx.x]);
}


// Grouped column sums: out_i[c] = sum_r x_i[r][c] for MANY small matrices in one launch (the per-scene partials of the bias /
// GroupNorm-affine / LayerNorm-gain gradients: one [scenes][n] matrix per layer, 80+ per training step).  blockIdx.y = item,
// blockIdx.x = 64-column slab; the block walks all rows (m is a few hundred), 8 loads in flight per thread.
struct ColsumItem { const float* x; long ldx; int m, n; float* out; };
__global__ __launch_bounds__(256) void colsum_grouped_kernel(const ColsumItem* __restrict__ items) {
    __shared__ float red[4][64];
    const ColsumItem it = items[blockIdx.y];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    if (blockIdx.x * 64 >= it.n) return;
    const int rq = threadIdx.x >> 6;
    float s = 0.f;
    if (c < it.n) {
        const float* xp = it.x + (long)rq * it.ldx + c;
        const long step = 4 * it.ldx;
        int r = rq;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (; r + 28 < it.m; r += 32, xp += 8 * step) {
            const float v0 = xp[0], v1 = xp[step], v2 = xp[2 * step], v3 = xp[3 * step];
            const float v4 = xp[4 * step], v5 = xp[5 * step], v6 = xp[6 * step], v7 = xp[7 * step];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
            s0 += v4; s1 += v5; s2 += v6; s3 += v7;
        }
        for (; r < it.m; r += 4, xp += step) s0 += xp[0];
        s = (s0 + s1) + (s2 + s3);
    }
    red[rq][threadIdx.x & 63] = s;
    __syncthreads();
    if (rq == 0 && c < it.n) it.out[c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// =====================================================================================================
// Backward of the fused GroupNorm + (scale+1, shift) + SiLU epilogue.  Block = (scene b, group g): 64 channels x N tokens.
//   z: pre-norm conv output (saved by the forward), dy: gradient of the block output.
//   dz = rstd * (dxh - mean(dxh) - xh * mean(dxh * xh)),  xh = (z - mu) * rstd, dxh = dgh * gamma,
//   dgh = du * (1 + scale), du = dy * silu'(u), u = (gamma * xh + beta) * (1 + scale) + shift.
// Per-channel token sums (dgamma, dbeta, dbias and, for per-scene conditioning, dscale/dshift) are written per scene
// and reduced over scenes by colsum_kernel: deterministic, no atomics.
// =====================================================================================================
struct GnBwdArgs {
    const float* z; long ldz;
    const float* dy; long ldy;
    const float* gamma; const float* beta;
    const float* ss; long ld_ss; int ss_mode;
    float* dz; long lddz;
    float* dgamma_p; float* dbeta_p; float* dbias_p;     // [scenes][...] rows pstride floats apart
    long pstride;
    float* dss; long ld_dss;                              // per-scene: [scenes][2C]; per-token/slot: [M][2C]
    int n_tok, C; float eps;
};

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Every thread owns one channel and every 4th token of the (scene, group) tile.  Its z / dy elements are read from HBM
// ONCE and parked in a thread-private LDS column (a register file spill area indexed by token: no barriers needed for
// it); the normalised activation and d(xhat) replace them after the first sweep, so the expensive SiLU derivative is
// evaluated once per element and the kernel moves exactly z + dy in, dz out.
__global__ __launch_bounds__(256) void gn_silu_bwd_kernel(const GnBwdArgs p) {
    extern __shared__ float gnb_cache[];          // [2][N][64]
    __shared__ float red[4];
    __shared__ float csum[5][4][64];
    const int b = blockIdx.x >> 3, g = blockIdx.x & 7;
    const int cl = threadIdx.x & 63, tq = threadIdx.x >> 6;
    const int c = g * 64 + cl;
    const int N = p.n_tok;
    const long tok0 = (long)b * N;
    const float inv_cnt = 1.0f / (64.0f * (float)N);
    float* zc = gnb_cache + cl;                   // zc[j * 64]
    float* dc = gnb_cache + (long)N * 64 + cl;
    // load sweep: both streams in flight together
    float s = 0.f;
    {
        const float* zp = p.z + (tok0 + tq) * p.ldz + c;
        const float* dp = p.dy + (tok0 + tq) * p.ldy + c;
        const long zs = 4 * p.ldz, ds = 4 * p.ldy;
#pragma unroll 5
        for (int j = tq; j < N; j += 4) {
            const float zv = *zp, dv = *dp;
            zp += zs; dp += ds;
            zc[j * 64] = zv; dc[j * 64] = dv;
            s += zv;
        }
    }
    // statistics of the group (two passes, as the forward)
    const float mu = block_sum(s, red) * inv_cnt;
    s = 0.f;
    for (int j = tq; j < N; j += 4) { const float d = zc[j * 64] - mu; s += d * d; }
    const float rs = 1.0f / sqrtf(block_sum(s, red) * inv_cnt + p.eps);
    const float ga = p.gamma[c], be = p.beta[c];

    const float* ssb = nullptr;                   // scale/shift row of token j: ssb + j * ss_step
    long ss_step = 0;
    if (p.ss) {
        if (p.ss_mode == DSC_SS_PER_TOKEN) { ssb = p.ss + tok0 * p.ld_ss; ss_step = p.ld_ss; }
        else if (p.ss_mode == DSC_SS_PER_SCENE) { ssb = p.ss + (long)b * p.ld_ss; ss_step = 0; }
        else if (p.ss_mode == DSC_SS_PER_SLOT) { ssb = p.ss; ss_step = p.ld_ss; }
    }
    const bool dss_rows = p.dss && p.ss_mode != DSC_SS_PER_SCENE && ssb;
    // sweep 1: d(xhat) of every element, group means of dxh and dxh*xh, per-channel sums
    float S1 = 0.f, S2 = 0.f, Gg = 0.f, Gb = 0.f, Gsc = 0.f, Gsh = 0.f;
    for (int j = tq; j < N; j += 4) {
        const float xh = (zc[j * 64] - mu) * rs;
        const float gh = ga * xh + be;
        float s1 = 1.0f, sh = 0.f;
        if (ssb) { const float* sr = ssb + (long)j * ss_step; s1 = 1.0f + sr[c]; sh = sr[p.C + c]; }
        const float u = gh * s1 + sh;
        const float sig = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
        const float du = dc[j * 64] * (sig * (1.0f + u * (1.0f - sig)));
        const float dgh = du * s1;
        const float dxh = dgh * ga;
        S1 += dxh; S2 += dxh * xh; Gg += dgh * xh; Gb += dgh; Gsc += du * gh; Gsh += du;
        if (dss_rows) {
            p.dss[(tok0 + j) * p.ld_dss + c] = du * gh;
            p.dss[(tok0 + j) * p.ld_dss + p.C + c] = du;
        }
        zc[j * 64] = xh;
        dc[j * 64] = dxh;
    }
    const float m1 = block_sum(S1, red) * inv_cnt;
    const float m2 = block_sum(S2, red) * inv_cnt;
    // sweep 2: dz and its per-channel sum (bias gradient)
    float Gz = 0.f;
    {
        float* op = p.dz + (tok0 + tq) * p.lddz + c;
        const long os = 4 * p.lddz;
        for (int j = tq; j < N; j += 4) {
            const float dzv = rs * (dc[j * 64] - m1 - zc[j * 64] * m2);
            *op = dzv;
            op += os;
            Gz += dzv;
        }
    }
    csum[0][tq][cl] = Gg; csum[1][tq][cl] = Gb; csum[2][tq][cl] = Gz; csum[3][tq][cl] = Gsc; csum[4][tq][cl] = Gsh;
    __syncthreads();
    if (tq == 0) {
        const long o = (long)b * p.pstride + c;
        p.dgamma_p[o] = (csum[0][0][cl] + csum[0][1][cl]) + (csum[0][2][cl] + csum[0][3][cl]);
        p.dbeta_p[o] = (csum[1][0][cl] + csum[1][1][cl]) + (csum[1][2][cl] + csum[1][3][cl]);
        p.dbias_p[o] = (csum[2][0][cl] + csum[2][1][cl]) + (csum[2][2][cl] + csum[2][3][cl]);
        if (p.dss && p.ss_mode == DSC_SS_PER_SCENE) {
            p.dss[(long)b * p.ld_dss + c] = (csum[3][0][cl] + csum[3][1][cl]) + (csum[3][2][cl] + csum[3][3][cl]);
            p.dss[(long)b * p.ld_dss + p.C + c] = (csum[4][0][cl] + csum[4][1][cl]) + (csum[4][2][cl] + csum[4][3][cl]);
        }
    }
}

// Register-resident form with 16-byte lanes (the LDS-cached 16-byte-lane kernel it replaced is in the git history): a thread keeps its NIT (token, channel quad) elements of z and dy in
// VGPRs (N <= 16 * NIT), so no LDS cache is needed (41 KB per block at N = 80 limited the round-1 kernel to 3 blocks per CU
// and cost four LDS passes); four block-wide reductions through an 8-entry double-buffered LDS slot (one barrier each).
template <int NIT>
__global__ __launch_bounds__(256) void gn_silu_bwd_reg_kernel(const GnBwdArgs p) {
    __shared__ float red[2][4];
    __shared__ float csum[5][4][64];
    const int b = blockIdx.x >> 3, g = blockIdx.x & 7;
    const int cq = threadIdx.x & 15, tr = threadIdx.x >> 4;                     // channel quad, token lane (0..15)
    const int wave = threadIdx.x >> 6;
    const int c = g * 64 + cq * 4;
    const int N = p.n_tok;
    const long tok0 = (long)b * N;
    const float inv_cnt = 1.0f / (64.0f * (float)N);
    int phase = 0;
    auto bsum = [&](float v) {                    // one barrier per reduction: alternate slots
        v = wave_sum(v);
        float* r = red[phase & 1];
        ++phase;
        if ((threadIdx.x & 63) == 0) r[wave] = v;
        __syncthreads();
        return (r[0] + r[1]) + (r[2] + r[3]);
    };
    f32x4 zv[NIT], dv[NIT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int j = tr + 16 * i;
        const bool ok = j < N;
        const long row = tok0 + (ok ? j : 0);
        zv[i] = *reinterpret_cast<const f32x4*>(p.z + row * p.ldz + c);
        dv[i] = *reinterpret_cast<const f32x4*>(p.dy + row * p.ldy + c);
        if (!ok) { zv[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i] = zv[i]; }
        s += (zv[i][0] + zv[i][1]) + (zv[i][2] + zv[i][3]);
    }
    const float mu = bsum(s) * inv_cnt;
    s = 0.f;
#pragma unroll
    for (int i = 0; i < NIT; ++i)
        if (tr + 16 * i < N) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = zv[i][e] - mu; s += d * d; }
        }
    const float rs = 1.0f / sqrtf(bsum(s) * inv_cnt + p.eps);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
    const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c);
    const float* ssb = nullptr;
    long ss_step = 0;
    if (p.ss) {
        if (p.ss_mode == DSC_SS_PER_TOKEN) { ssb = p.ss + tok0 * p.ld_ss; ss_step = p.ld_ss; }
        else if (p.ss_mode == DSC_SS_PER_SCENE) { ssb = p.ss + (long)b * p.ld_ss; ss_step = 0; }
        else if (p.ss_mode == DSC_SS_PER_SLOT) { ssb = p.ss; ss_step = p.ld_ss; }
    }
    const bool dss_rows = p.dss && p.ss_mode != DSC_SS_PER_SCENE && ssb;
    float S1 = 0.f, S2 = 0.f;
    f32x4 Gg = {0.f, 0.f, 0.f, 0.f}, Gb = Gg, Gsc = Gg, Gsh = Gg, Gz = Gg;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int j = tr + 16 * i;
        if (j < N) {
            f32x4 s1 = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (ssb) {
                const float* sr = ssb + (long)j * ss_step;
                const f32x4 a = *reinterpret_cast<const f32x4*>(sr + c);
                sh = *reinterpret_cast<const f32x4*>(sr + p.C + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) s1[e] = 1.0f + a[e];
            }
            f32x4 xh, dxh, d_sc, d_sh;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[e] = (zv[i][e] - mu) * rs;
                const float gh = ga[e] * xh[e] + be[e];
                const float u = gh * s1[e] + sh[e];
                const float sig = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
                const float du = dv[i][e] * (sig * (1.0f + u * (1.0f - sig)));
                const float dgh = du * s1[e];
                dxh[e] = dgh * ga[e];
                S1 += dxh[e]; S2 += dxh[e] * xh[e];
                Gg[e] += dgh * xh[e]; Gb[e] += dgh; Gsc[e] += du * gh; Gsh[e] += du;
                d_sc[e] = du * gh; d_sh[e] = du;
            }
            if (dss_rows) {
                *reinterpret_cast<f32x4*>(p.dss + (tok0 + j) * p.ld_dss + c) = d_sc;
                *reinterpret_cast<f32x4*>(p.dss + (tok0 + j) * p.ld_dss + p.C + c) = d_sh;
            }
            zv[i] = xh;
            dv[i] = dxh;
        }
    }
    const float m1 = bsum(S1) * inv_cnt;
    const float m2 = bsum(S2) * inv_cnt;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int j = tr + 16 * i;
        if (j < N) {
            f32x4 dzv;
#pragma unroll
            for (int e = 0; e < 4; ++e) { dzv[e] = rs * (dv[i][e] - m1 - zv[i][e] * m2); Gz[e] += dzv[e]; }
            *reinterpret_cast<f32x4*>(p.dz + (tok0 + j) * p.lddz + c) = dzv;
        }
    }
    auto fold = [&](f32x4 v, int slot) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = v[e];
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            if ((threadIdx.x & 63) < 16) csum[slot][wave][cq * 4 + e] = t;
        }
    };
    fold(Gg, 0); fold(Gb, 1); fold(Gz, 2); fold(Gsc, 3); fold(Gsh, 4);
    __syncthreads();
    if (threadIdx.x < 64) {
        const int cl = threadIdx.x;
        const long o = (long)b * p.pstride + g * 64 + cl;
        p.dgamma_p[o] = (csum[0][0][cl] + csum[0][1][cl]) + (csum[0][2][cl] + csum[0][3][cl]);
        p.dbeta_p[o] = (csum[1][0][cl] + csum[1][1][cl]) + (csum[1][2][cl] + csum[1][3][cl]);
        p.dbias_p[o] = (csum[2][0][cl] + csum[2][1][cl]) + (csum[2][2][cl] + csum[2][3][cl]);
        if (p.dss && p.ss_mode == DSC_SS_PER_SCENE) {
            p.dss[(long)b * p.ld_dss + g * 64 + cl] = (csum[3][0][cl] + csum[3][1][cl]) + (csum[3][2][cl] + csum[3][3][cl]);
            p.dss[(long)b * p.ld_dss + p.C + g * 64 + cl] = (csum[4][0][cl] + csum[4][1][cl]) + (csum[4][2][cl] + csum[4][3][cl]);
        }
    }
}

// =====================================================================================================
// weight standardisation backward, one block per weight row (batched like the forward)
// =====================================================================================================
struct WsBwdItem { const float* w; const float* dwh; float* dw; int rows; int cols; };
struct WsBwdBatch { WsBwdItem it[DSC_WS_MAX]; };

__global__ __launch_bounds__(256) void ws_bwd_kernel(const WsBwdBatch bch, const float eps) {
    __shared__ float red[4];
    const WsBwdItem it = bch.it[blockIdx.y];
    const int row = blockIdx.x;
    if (row >= it.rows) return;
    const float* w = it.w + (long)row * it.cols;
    const float* g = it.dwh + (long)row * it.cols;
    float* o = it.dw + (long)row * it.cols;
    constexpr int MAXE = 8;
    float v[MAXE], gv[MAXE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int c = threadIdx.x + 256 * i;
        v[i] = (c < it.cols) ? w[c] : 0.f;
        gv[i] = (c < it.cols) ? g[c] : 0.f;
        s += v[i];
    }
    const float inv = 1.0f / (float)it.cols;
    const float mean = block_sum(s, red) * inv;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int c = threadIdx.x + 256 * i;
        const float d = (c < it.cols) ? v[i] - mean : 0.f;
        s2 += d * d;
    }
    const float rs = 1.0f / sqrtf(block_sum(s2, red) * inv + eps);
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < it.cols) { v[i] = (v[i] - mean) * rs; a1 += gv[i]; a2 += gv[i] * v[i]; }
    }
    const float m1 = block_sum(a1, red) * inv;
    const float m2 = block_sum(a2, red) * inv;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < it.cols) o[c] = rs * (gv[i] - m1 - v[i] * m2);
    }
}

// =====================================================================================================
// channel LayerNorm backward (d = 512): wave per row, block-strided over rows, gain-gradient partial per block
// =====================================================================================================
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, long ldx,
                                                           const float* __restrict__ g,
                                                           const float* __restrict__ dy, long ldy,
                                                           float* __restrict__ dx, long lddx,
                                                           const float* __restrict__ addend, long ldadd,
                                                           float* __restrict__ dg_partial, int m, float eps) {
    __shared__ float red[4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 ga = *reinterpret_cast<const f32x4*>(g + lane * 4);
    const f32x4 gb = *reinterpret_cast<const f32x4*>(g + 256 + lane * 4);
    float dga[4] = {0.f, 0.f, 0.f, 0.f}, dgb[4] = {0.f, 0.f, 0.f, 0.f};
    for (long row = (long)blockIdx.x * 4 + wave; row < m; row += (long)gridDim.x * 4) {
        const float* xr = x + row * ldx;
        const float* dr = dy + row * ldy;
        const f32x4 a = *reinterpret_cast<const f32x4*>(xr + lane * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(xr + 256 + lane * 4);
        const f32x4 da = *reinterpret_cast<const f32x4*>(dr + lane * 4);
        const f32x4 db = *reinterpret_cast<const f32x4*>(dr + 256 + lane * 4);
        float s = (a[0] + a[1]) + (a[2] + a[3]) + (b[0] + b[1]) + (b[2] + b[3]);
        const float mean = wave_sum(s) * (1.0f / 512.0f);
        float s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float u = a[e] - mean, w = b[e] - mean; s2 += u * u + w * w; }
        const float rs = 1.0f / sqrtf(wave_sum(s2) * (1.0f / 512.0f) + eps);
        float xa[4], xb[4], ha[4], hb[4];
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            xa[e] = (a[e] - mean) * rs; xb[e] = (b[e] - mean) * rs;
            ha[e] = da[e] * ga[e]; hb[e] = db[e] * gb[e];
            t1 += ha[e] + hb[e];
            t2 += ha[e] * xa[e] + hb[e] * xb[e];
            dga[e] += da[e] * xa[e]; dgb[e] += db[e] * xb[e];
        }
        const float m1 = wave_sum(t1) * (1.0f / 512.0f);
        const float m2 = wave_sum(t2) * (1.0f / 512.0f);
        f32x4 oa, ob;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            oa[e] = rs * (ha[e] - m1 - xa[e] * m2);
            ob[e] = rs * (hb[e] - m1 - xb[e] * m2);
        }
        if (addend) {                              // dx = addend + d LayerNorm: the gradient x already holds from another consumer
            const float* ar = addend + row * ldadd;
            oa += *reinterpret_cast<const f32x4*>(ar + lane * 4);
            ob += *reinterpret_cast<const f32x4*>(ar + 256 + lane * 4);
        }
        float* o = dx + row * lddx;
        *reinterpret_cast<f32x4*>(o + lane * 4) = oa;
        *reinterpret_cast<f32x4*>(o + 256 + lane * 4) = ob;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[wave][lane * 4 + e] = dga[e]; red[wave][256 + lane * 4 + e] = dgb[e]; }
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256)
        dg_partial[(long)blockIdx.x * 512 + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// =====================================================================================================
// linear attention backward, 256-thread block = (scene, head).  Dynamic LDS (rows padded to 36 floats, float4 access):
//   Ks (softmaxed k), Vs, DK [nk][36]; Qs (scale * softmax q), DO [nq][36]; ctx, dctx [32][36]; cvec [32]
//   phase 1  k softmax over tokens (8 lanes / channel), q softmax over channels (4 lanes / token)
//   phase 2  ctx[d][e] = sum_j ks[j][d] v[j][e],  dctx[d][e] = sum_i qs[i][d] dout[i][e]          (4 outputs / thread)
//   phase 3  dq  (4 lanes / query token),  phase 4  dks, dv (4 lanes / key token),  phase 5  dk = ks * (dks - <dks, ks>)
// =====================================================================================================
constexpr int HP = 33;      // softmax-attention backward below
constexpr int LP = 36;

__global__ __launch_bounds__(256) void linear_attention_bwd_kernel(
        const float* __restrict__ q, long ldq, const float* __restrict__ k, long ldk, const float* __restrict__ v, long ldv,
        const float* __restrict__ dout, long ldo, float* __restrict__ dq, long lddq, float* __restrict__ dk, long lddk,
        float* __restrict__ dv, long lddv, int nq, int nk, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ks = lds;
    float* Vs = Ks + nk * LP;
    float* DK = Vs + nk * LP;
    float* Qs = DK + nk * LP;
    float* DO = Qs + nq * LP;
    float* ctx = DO + nq * LP;
    float* dctx = ctx + 32 * LP;
    float* cvec = dctx + 32 * LP;
    const int b = blockIdx.x >> 2, h = blockIdx.x & 3;
    const int tid = threadIdx.x;
    for (int f = tid; f < nk * 8; f += 256) {
        const int j = f >> 3, c4 = (f & 7) * 4;
        const long row = (long)b * nk + j;
        *reinterpret_cast<f32x4*>(Ks + j * LP + c4) = *reinterpret_cast<const f32x4*>(k + row * ldk + h * 32 + c4);
        *reinterpret_cast<f32x4*>(Vs + j * LP + c4) = *reinterpret_cast<const f32x4*>(v + row * ldv + h * 32 + c4);
    }
    for (int f = tid; f < nq * 8; f += 256) {
        const int i = f >> 3, c4 = (f & 7) * 4;
        const long row = (long)b * nq + i;
        *reinterpret_cast<f32x4*>(Qs + i * LP + c4) = *reinterpret_cast<const f32x4*>(q + row * ldq + h * 32 + c4);
        *reinterpret_cast<f32x4*>(DO + i * LP + c4) = *reinterpret_cast<const f32x4*>(dout + row * ldo + h * 32 + c4);
    }
    __syncthreads();
    {   // phase 1a: ks = softmax over tokens (normalised in place)
        const int d = tid >> 3, part = tid & 7;
        float mx = -INFINITY;
        for (int j = part; j < nk; j += 8) mx = fmaxf(mx, Ks[j * LP + d]);
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
        float sm = 0.f;
        for (int j = part; j < nk; j += 8) { const float e = expf(Ks[j * LP + d] - mx); Ks[j * LP + d] = e; sm += e; }
        sm += __shfl_xor(sm, 1, 64);
        sm += __shfl_xor(sm, 2, 64);
        sm += __shfl_xor(sm, 4, 64);
        const float inv = 1.0f / sm;
        for (int j = part; j < nk; j += 8) Ks[j * LP + d] *= inv;
    }
    // phase 1b: qs = scale * softmax over channels, 4 lanes per token (each lane touches only its own 8 channels)
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
            *reinterpret_cast<f32x4*>(Qs + i * LP + part * 8) = a;
            *reinterpret_cast<f32x4*>(Qs + i * LP + part * 8 + 4) = c;
        }
    }
    __syncthreads();
    {   // phase 2: thread -> (d, 4 consecutive e) of ctx and dctx
        const int d = tid >> 3, e0 = (tid & 7) * 4;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < nk; ++j) {
            const float kd = Ks[j * LP + d];
            const f32x4 vv = *reinterpret_cast<const f32x4*>(Vs + j * LP + e0);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += kd * vv[e];
        }
        for (int i = 0; i < nq; ++i) {
            const float qd = Qs[i * LP + d];
            const f32x4 dd = *reinterpret_cast<const f32x4*>(DO + i * LP + e0);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] += qd * dd[e];
        }
        *reinterpret_cast<f32x4*>(ctx + d * LP + e0) = a;
        *reinterpret_cast<f32x4*>(dctx + d * LP + e0) = g;
    }
    __syncthreads();
    // phase 3: dq.  dqs[d] = sum_e dout[i][e] ctx[d][e];  dq[d] = qs[d] * (dqs[d] - sum_d' dqs[d'] qs[d'] / scale)
    for (int i0 = 0; i0 < nq; i0 += 64) {
        const int i = i0 + (tid >> 2), part = tid & 3;
        const bool ok = i < nq;
        float dqs[8];
        float dot = 0.f;
        if (ok) {
            float dov[32];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(DO + i * LP + c * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) dov[c * 4 + e] = t4[e];
            }
#pragma unroll
            for (int dd = 0; dd < 8; ++dd) {
                const int d = part * 8 + dd;
                float sdd = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(ctx + d * LP + c * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) sdd += dov[c * 4 + e] * t4[e];
                }
                dqs[dd] = sdd;
                dot += sdd * Qs[i * LP + d];
            }
        }
        dot += __shfl_xor(dot, 1, 64);
        dot += __shfl_xor(dot, 2, 64);
        if (ok) {
            const float corr = dot / scale;
            f32x4 a, c;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = Qs[i * LP + part * 8 + e] * (dqs[e] - corr);
                c[e] = Qs[i * LP + part * 8 + 4 + e] * (dqs[4 + e] - corr);
            }
            float* o = dq + ((long)b * nq + i) * lddq + h * 32 + part * 8;
            *reinterpret_cast<f32x4*>(o) = a;
            *reinterpret_cast<f32x4*>(o + 4) = c;
        }
    }
    // phase 4: per key token (4 lanes): dks for its 8 channels, dv for its 8 outputs
    for (int j0 = 0; j0 < nk; j0 += 64) {
        const int j = j0 + (tid >> 2), part = tid & 3;
        if (j < nk) {
            float vv[32];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(Vs + j * LP + c * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) vv[c * 4 + e] = t4[e];
            }
            f32x4 k0, k1;
#pragma unroll
            for (int dd = 0; dd < 8; ++dd) {
                const int d = part * 8 + dd;
                float sdd = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(dctx + d * LP + c * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) sdd += vv[c * 4 + e] * t4[e];
                }
                if (dd < 4) k0[dd] = sdd; else k1[dd - 4] = sdd;
            }
            *reinterpret_cast<f32x4*>(DK + j * LP + part * 8) = k0;
            *reinterpret_cast<f32x4*>(DK + j * LP + part * 8 + 4) = k1;
            f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int d = 0; d < 32; ++d) {
                const float kd = Ks[j * LP + d];
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(dctx + d * LP + part * 8);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(dctx + d * LP + part * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { o0[e] += kd * c0[e]; o1[e] += kd * c1[e]; }
            }
            float* o = dv + ((long)b * nk + j) * lddv + h * 32 + part * 8;
            *reinterpret_cast<f32x4*>(o) = o0;
            *reinterpret_cast<f32x4*>(o + 4) = o1;
        }
    }
    __syncthreads();
    {   // phase 5a: per channel <dks, ks> over the tokens
        const int d = tid >> 3, part = tid & 7;
        float sdd = 0.f;
        for (int j = part; j < nk; j += 8) sdd += DK[j * LP + d] * Ks[j * LP + d];
        sdd += __shfl_xor(sdd, 1, 64);
        sdd += __shfl_xor(sdd, 2, 64);
        sdd += __shfl_xor(sdd, 4, 64);
        if (part == 0) cvec[d] = sdd;
    }
    __syncthreads();
    for (int f = tid; f < nk * 8; f += 256) {
        const int j = f >> 3, c4 = (f & 7) * 4;
        const f32x4 ks4 = *reinterpret_cast<const f32x4*>(Ks + j * LP + c4);
        const f32x4 dk4 = *reinterpret_cast<const f32x4*>(DK + j * LP + c4);
        f32x4 t4;
#pragma unroll
        for (int e = 0; e < 4; ++e) t4[e] = ks4[e] * (dk4[e] - cvec[c4 + e]);
        *reinterpret_cast<f32x4*>(dk + ((long)b * nk + j) * lddk + h * 32 + c4) = t4;
    }
}

// =====================================================================================================
// softmax attention backward, block = (scene, head); phase 1 thread per query, phase 2 thread per key (recompute p)
// dynamic LDS: Qs (scaled), Ks, Vs, DO [n][33]; row stats mx, inv_sum, D [n]
// =====================================================================================================
__global__ __launch_bounds__(192) void attention_bwd_kernel(
        const float* __restrict__ q, long ldq, const float* __restrict__ k, long ldk, const float* __restrict__ v, long ldv,
        const float* __restrict__ dout, long ldo, float* __restrict__ dq, long lddq, float* __restrict__ dk, long lddk,
        float* __restrict__ dv, long lddv, int n, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Qs = lds;
    float* Ks = Qs + n * HP;
    float* Vs = Ks + n * HP;
    float* DO = Vs + n * HP;
    float* rmx = DO + n * HP;
    float* rinv = rmx + n;
    float* rD = rinv + n;
    const int b = blockIdx.x >> 2, h = blockIdx.x & 3;
    const int tid = threadIdx.x;
    for (int f = tid; f < n * 8; f += 192) {
        const int j = f >> 3, c4 = (f & 7) * 4;
        const long row = (long)b * n + j;
        const f32x4 qv = *reinterpret_cast<const f32x4*>(q + row * ldq + h * 32 + c4);
        const f32x4 kv = *reinterpret_cast<const f32x4*>(k + row * ldk + h * 32 + c4);
        const f32x4 vv = *reinterpret_cast<const f32x4*>(v + row * ldv + h * 32 + c4);
        const f32x4 dv4 = *reinterpret_cast<const f32x4*>(dout + row * ldo + h * 32 + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Qs[j * HP + c4 + e] = qv[e] * scale; Ks[j * HP + c4 + e] = kv[e];
            Vs[j * HP + c4 + e] = vv[e]; DO[j * HP + c4 + e] = dv4[e];
        }
    }
    __syncthreads();
    if (tid < n) {
        const int i = tid;
        float qv[32], dov[32];
#pragma unroll
        for (int d = 0; d < 32; ++d) { qv[d] = Qs[i * HP + d]; dov[d] = DO[i * HP + d]; }
        float mx = -INFINITY;
        for (int j = 0; j < n; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) s += qv[d] * Ks[j * HP + d];
            mx = fmaxf(mx, s);
        }
        float sm = 0.f, Dn = 0.f;
        for (int j = 0; j < n; ++j) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) { s += qv[d] * Ks[j * HP + d]; dp += dov[d] * Vs[j * HP + d]; }
            const float pj = expf(s - mx);
            sm += pj;
            Dn += pj * dp;
        }
        const float inv = 1.0f / sm;
        const float Di = Dn * inv;               // sum_j p_ij dP_ij
        rmx[i] = mx; rinv[i] = inv; rD[i] = Di;
        float dqv[32];
#pragma unroll
        for (int d = 0; d < 32; ++d) dqv[d] = 0.f;
        for (int j = 0; j < n; ++j) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) { s += qv[d] * Ks[j * HP + d]; dp += dov[d] * Vs[j * HP + d]; }
            const float ds = expf(s - mx) * inv * (dp - Di);
#pragma unroll
            for (int d = 0; d < 32; ++d) dqv[d] += ds * Ks[j * HP + d];
        }
        float* o = dq + ((long)b * n + i) * lddq + h * 32;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            f32x4 t4;
#pragma unroll
            for (int e = 0; e < 4; ++e) t4[e] = dqv[c * 4 + e] * scale;
            *reinterpret_cast<f32x4*>(o + c * 4) = t4;
        }
    }
    __syncthreads();
    if (tid < n) {
        const int j = tid;
        float kv[32], vv[32], dkv[32], dvv[32];
#pragma unroll
        for (int d = 0; d < 32; ++d) { kv[d] = Ks[j * HP + d]; vv[d] = Vs[j * HP + d]; dkv[d] = 0.f; dvv[d] = 0.f; }
        for (int i = 0; i < n; ++i) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) { s += Qs[i * HP + d] * kv[d]; dp += DO[i * HP + d] * vv[d]; }
            const float pij = expf(s - rmx[i]) * rinv[i];
            const float ds = pij * (dp - rD[i]);
#pragma unroll
            for (int d = 0; d < 32; ++d) { dkv[d] += ds * Qs[i * HP + d]; dvv[d] += pij * DO[i * HP + d]; }
        }
        float* ok = dk + ((long)b * n + j) * lddk + h * 32;
        float* ov = dv + ((long)b * n + j) * lddv + h * 32;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            f32x4 a4, b4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a4[e] = dkv[c * 4 + e]; b4[e] = dvv[c * 4 + e]; }
            *reinterpret_cast<f32x4*>(ok + c * 4) = a4;
            *reinterpret_cast<f32x4*>(ov + c * 4) = b4;
        }
    }
}

// Same backward for scenes of up to 96 tokens: the probabilities P and dS = P (dP - D) are kept in LDS instead of being
// recomputed per key, and 4 lanes share a query / key (8 of the 32 head channels each, 16-byte LDS reads, quad shuffles).
// LDS: Q (pre-scaled), K, V, dO as [n][36]; P, dS as [n][n+1].
__global__ __launch_bounds__(512) void attention_bwd_cached_kernel(
        const float* __restrict__ q, long ldq, const float* __restrict__ k, long ldk, const float* __restrict__ v, long ldv,
        const float* __restrict__ dout, long ldo, float* __restrict__ dq, long lddq, float* __restrict__ dk, long lddk,
        float* __restrict__ dv, long lddv, int n, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LQ = 36;
    const int PS = n + 1;
    float* Qs = lds;
    float* Ks = Qs + n * LQ;
    float* Vs = Ks + n * LQ;
    float* DO = Vs + n * LQ;
    float* P = DO + n * LQ;
    float* DS = P + n * PS;
    const int b = blockIdx.x >> 2, h = blockIdx.x & 3;
    const int tid = threadIdx.x, part = tid & 3;
    const int nthr = blockDim.x, rows = nthr >> 2;       // four lanes per token (dsc_attention_threads)
    for (int f = tid; f < n * 8; f += nthr) {
        const int j = f >> 3, c4 = (f & 7) * 4;
        const long row = (long)b * n + j;
        f32x4 qv = *reinterpret_cast<const f32x4*>(q + row * ldq + h * 32 + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) qv[e] *= scale;
        *reinterpret_cast<f32x4*>(Qs + j * LQ + c4) = qv;
        *reinterpret_cast<f32x4*>(Ks + j * LQ + c4) = *reinterpret_cast<const f32x4*>(k + row * ldk + h * 32 + c4);
        *reinterpret_cast<f32x4*>(Vs + j * LQ + c4) = *reinterpret_cast<const f32x4*>(v + row * ldv + h * 32 + c4);
        *reinterpret_cast<f32x4*>(DO + j * LQ + c4) = *reinterpret_cast<const f32x4*>(dout + row * ldo + h * 32 + c4);
    }
    __syncthreads();
    auto dot8 = [&](const float* a, const float (&x)[8]) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(a), a1 = *reinterpret_cast<const f32x4*>(a + 4);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) s += x[e] * a0[e];
#pragma unroll
        for (int e = 0; e < 4; ++e) s += x[4 + e] * a1[e];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        return s;
    };
    // phase 1: per query i -- P[i][:], D_i, dS[i][:], dq_i
    for (int i0 = 0; i0 < n; i0 += rows) {
        const int i = i0 + (tid >> 2);
        const bool ok = i < n;
        const int ic = ok ? i : 0;
        float qv[8], dov[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { qv[e] = Qs[ic * LQ + part * 8 + e]; dov[e] = DO[ic * LQ + part * 8 + e]; }
        float mx = -INFINITY;
        for (int j = 0; j < n; ++j) {
            const float s = dot8(Ks + j * LQ + part * 8, qv);
            if (ok && part == 0) P[i * PS + j] = s;
            mx = fmaxf(mx, s);
        }
        float sm = 0.f, Dn = 0.f;
        for (int j = 0; j < n; ++j) {
            const float dp = dot8(Vs + j * LQ + part * 8, dov);
            const float pj = expf((ok ? P[i * PS + j] : 0.f) - mx);      // own write (part 0) visible: same wave, in order
            sm += pj;
            Dn += pj * dp;
            if (ok && part == 0) { P[i * PS + j] = pj; DS[i * PS + j] = dp; }
        }
        const float inv = 1.0f / sm;
        const float Di = Dn * inv;
        float dqv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < n; ++j) {
            const float pij = (ok ? P[i * PS + j] : 0.f) * inv;
            const float ds = pij * ((ok ? DS[i * PS + j] : 0.f) - Di);
            if (ok && part == 0) { P[i * PS + j] = pij; DS[i * PS + j] = ds; }
            const f32x4 k0 = *reinterpret_cast<const f32x4*>(Ks + j * LQ + part * 8);
            const f32x4 k1 = *reinterpret_cast<const f32x4*>(Ks + j * LQ + part * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { dqv[e] += ds * k0[e]; dqv[4 + e] += ds * k1[e]; }
        }
        if (ok) {
            float* o = dq + ((long)b * n + i) * lddq + h * 32 + part * 8;
            f32x4 t0, t1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { t0[e] = dqv[e] * scale; t1[e] = dqv[4 + e] * scale; }
            *reinterpret_cast<f32x4*>(o) = t0;
            *reinterpret_cast<f32x4*>(o + 4) = t1;
        }
    }
    __syncthreads();
    // phase 2: per key j -- dv_j = sum_i P_ij dO_i, dk_j = sum_i dS_ij Q_i (Q already carries the scale)
    for (int j0 = 0; j0 < n; j0 += rows) {
        const int j = j0 + (tid >> 2);
        const bool ok = j < n;
        const int jc = ok ? j : 0;
        float dkv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dvv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < n; ++i) {
            const float pij = P[i * PS + jc], ds = DS[i * PS + jc];
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(Qs + i * LQ + part * 8);
            const f32x4 q1 = *reinterpret_cast<const f32x4*>(Qs + i * LQ + part * 8 + 4);
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(DO + i * LQ + part * 8);
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(DO + i * LQ + part * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dkv[e] += ds * q0[e]; dkv[4 + e] += ds * q1[e];
                dvv[e] += pij * d0[e]; dvv[4 + e] += pij * d1[e];
            }
        }
        if (ok) {
            float* ok_ = dk + ((long)b * n + j) * lddk + h * 32 + part * 8;
            float* ov = dv + ((long)b * n + j) * lddv + h * 32 + part * 8;
            f32x4 a0, a1, b0, b1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a0[e] = dkv[e]; a1[e] = dkv[4 + e]; b0[e] = dvv[e]; b1[e] = dvv[4 + e]; }
            *reinterpret_cast<f32x4*>(ok_) = a0; *reinterpret_cast<f32x4*>(ok_ + 4) = a1;
            *reinterpret_cast<f32x4*>(ov) = b0; *reinterpret_cast<f32x4*>(ov + 4) = b1;
        }
    }
}

// dx = dy * act'(x)
__global__ void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                               long count, int act) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < count; i += stride) {
        dx[i] = dy[i] * dsc_act_grad(x[i], act);
    }
}

// out[c][r] = in[r][c], 32x32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, long ldi, float* __restrict__ out,
                                                       long ldo, int rows, int cols) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (r0 + r < rows && c0 + tx < cols) tile[r][tx] = in[(long)(r0 + r) * ldi + c0 + tx];
    __syncthreads();
    for (int c = ty; c < 32; c += 8)
        if (c0 + c < cols && r0 + tx < rows) out[(long)(c0 + c) * ldo + r0 + tx] = tile[tx][c];
}

// batched form: one launch transposes every matrix of a list (blockIdx.z = matrix); the training backward needs W^T of all
// 56 standardised conv weights, produced right after the standardisation instead of one launch per layer
struct TransposeBatch { dsc_ws_item it[DSC_WS_MAX]; };

__global__ __launch_bounds__(256) void transpose_batched_kernel(const TransposeBatch bch) {
    __shared__ float tile[32][33];
    const dsc_ws_item it = bch.it[blockIdx.z];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    if (c0 >= it.cols || r0 >= it.rows) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (r0 + r < it.rows && c0 + tx < it.cols) tile[r][tx] = it.w[(long)(r0 + r) * it.cols + c0 + tx];
    __syncthreads();
    for (int c = ty; c < 32; c += 8)
        if (c0 + c < it.cols && r0 + tx < it.rows) it.out[(long)(c0 + c) * it.rows + r0 + tx] = tile[tx][c];
}

// 16-byte-lane form: 64 x 64 tiles, a wave reads 4 rows x 256 B and writes 4 rows x 256 B per instruction (the scalar form
// above moves 128-byte segments 4 B per lane: ~1 TB/s on the 600 MB of weights transposed per training step).  Needs rows and
// columns that are multiples of 4 and 16-byte aligned matrices.
__global__ __launch_bounds__(256) void transpose_batched_v4_kernel(const TransposeBatch bch) {
    __shared__ float tile[64][65];
    const dsc_ws_item it = bch.it[blockIdx.z];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    if (c0 >= it.cols || r0 >= it.rows) return;
    const int q = threadIdx.x & 15, rr = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = rr + 16 * i;
        if (r0 + r < it.rows && c0 + q * 4 < it.cols) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(it.w + (long)(r0 + r) * it.cols + c0 + q * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[r][q * 4 + e] = v[e];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = rr + 16 * i;                                  // output row = input column
        if (c0 + c < it.cols && r0 + q * 4 < it.rows) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = tile[q * 4 + e][c];
            *reinterpret_cast<f32x4*>(it.out + (long)(c0 + c) * it.rows + r0 + q * 4) = v;
        }
    }
}

}  // namespace

// --------------------------------------------------------------------------------------------- C ABI
static void tn_split(int m, int n, int k, int* splits, int* chunk) {
    const int ktiles = (k + 127) / 128, ntiles = (n + 127) / 128;
    int s = (512 + ktiles * ntiles - 1) / (ktiles * ntiles);        // aim at ~2 blocks per CU
    const int maxs = (m + 31) / 32;
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    *chunk = ((m + s - 1) / s + 31) / 32 * 32;
    *splits = (m + *chunk - 1) / *chunk;
}

extern "C" int dsc_gemm_tn_f32(const float* a1, int64_t lda1, int32_t k1, const float* a2, int64_t lda2, int32_t k2,
                               const float* dy, int64_t ldd, float* out, int64_t ldo, float* dbias, int32_t m, int32_t n,
                               int32_t kvalid, float* workspace, int64_t workspace_floats, dsc_stream_t stream) {
    if (!a1 || !dy || !out || m < 1 || n < 1 || k1 < 1 || k2 < 0 || (k2 > 0 && !a2)) return DSC_EINVAL;
    if ((k1 & 3) || (k2 & 3) || (n & 3)) return DSC_EINVAL;
    if (k2 > 0 && (k1 % 128)) return DSC_EINVAL;                       // a k tile never straddles the segments
    if (!dsc_aligned16(a1) || (lda1 & 3) || !dsc_aligned16(dy) || (ldd & 3)) return DSC_EALIGN;
    if (k2 > 0 && (!dsc_aligned16(a2) || (lda2 & 3))) return DSC_EALIGN;
    const int K = k1 + k2;
    if (kvalid < 1 || kvalid > K) return DSC_EINVAL;
    const int ktiles = (K + 127) / 128, ntiles = (n + 127) / 128;
    int splits, chunk;
    tn_split(m, n, K, &splits, &chunk);
    const long wslab = (long)n * ldo;                                  // one weight-gradient slab
    hipStream_t s = static_cast<hipStream_t>(stream);
    TnArgs p{a1, (long)lda1, k1, a2, (long)lda2, k2, dy, (long)ldd, out, (long)ldo, dbias, 0, 0, m, n, kvalid, chunk, ktiles};
    float* bias_ws = nullptr;
    if (splits > 1) {
        if (!workspace || workspace_floats < (wslab + n) * splits) return DSC_EINVAL;
        if (ldo != kvalid) return DSC_EINVAL;                          // slab reduction assumes a dense [n][kvalid] output
        bias_ws = workspace + wslab * splits;                          // compact [splits][n] area behind the weight slabs
        p.out = workspace;
        p.bias_out = dbias ? bias_ws : nullptr;
        p.slab = wslab;
        p.bias_slab = n;
    }
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(ktiles * ntiles, splits), dim3(256), 0, s, p);
    DSC_LAUNCH_CHECK();
    if (splits > 1) {
        long blocks = ((wslab & 3) == 0) ? (wslab / 4 + 255) / 256 : (wslab + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        DSC_CLEAR_STALE_ERROR();
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, s, workspace, wslab, splits, out, wslab);
        DSC_LAUNCH_CHECK();
        if (dbias) {
            DSC_CLEAR_STALE_ERROR();
            hipLaunchKernelGGL(reduce_slabs_kernel, dim3((n + 255) / 256), dim3(256), 0, s, bias_ws, (long)n, splits,
                               dbias, (long)n);
            DSC_LAUNCH_CHECK();
        }
    }
    return 0;
}

// slab reduction of a grouped launch (shared with gemm_tn_split.hip, whose 256 x 128 tiles write the same slab layout)
int dsc_launch_reduce_grouped(const dsc_tn_group* groups_dev, int count, int total_tiles, int splits, const float* workspace,
                              hipStream_t s) {
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(reduce_grouped_kernel, dim3((unsigned)total_tiles), dim3(256), 0, s, groups_dev, count, splits, workspace);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_gemm_tn_grouped_f32(const dsc_tn_group* groups_dev, int32_t count, int32_t total_tiles, int32_t splits,
                                       float* workspace, int64_t workspace_floats, int64_t workspace_needed,
                                       dsc_stream_t stream) {
    if (!groups_dev || count < 1 || total_tiles < count || splits < 1 || splits > 64) return DSC_EINVAL;
    if (splits > 1 && (!workspace || workspace_floats < workspace_needed || workspace_needed < 1)) return DSC_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3((unsigned)total_tiles, (unsigned)splits), dim3(256), 0, s, groups_dev, count,
                       splits, workspace);
    DSC_LAUNCH_CHECK();
    if (splits > 1) {
        DSC_CLEAR_STALE_ERROR();
        hipLaunchKernelGGL(reduce_grouped_kernel, dim3((unsigned)total_tiles), dim3(256), 0, s, groups_dev, count, splits,
                           workspace);
        DSC_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int64_t dsc_gemm_tn_workspace_floats(int32_t m, int32_t n, int32_t k) {
    int splits, chunk;
    tn_split(m, n, k, &splits, &chunk);
    return splits > 1 ? ((int64_t)n * k + n) * splits : 0;
}

extern "C" int dsc_colsum_f32(const float* x, int64_t ldx, int32_t m, int32_t n, float* out, float* workspace,
                              int64_t workspace_floats, dsc_stream_t stream) {
    if (!x || !out || m < 1 || n < 1) return DSC_EINVAL;
    int splits = (m + 255) / 256;
    if (splits > 64) splits = 64;
    const int chunk = (m + splits - 1) / splits;
    splits = (m + chunk - 1) / chunk;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (splits == 1) {
        DSC_CLEAR_STALE_ERROR();
        hipLaunchKernelGGL(colsum_kernel, dim3((n + 63) / 64, 1), dim3(256), 0, s, x, (long)ldx, m, n, chunk, out);
        DSC_LAUNCH_CHECK();
        return 0;
    }
    if (!workspace || workspace_floats < (int64_t)splits * n) return DSC_EINVAL;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(colsum_kernel, dim3((n + 63) / 64, splits), dim3(256), 0, s, x, (long)ldx, m, n, chunk, workspace);
    DSC_LAUNCH_CHECK();
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((n + 255) / 256), dim3(256), 0, s, workspace, (long)n, splits, out, (long)n);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_colsum_grouped_f32(const dsc_colsum_item* items_dev, int32_t count, int32_t max_n, dsc_stream_t stream) {
    if (!items_dev || count < 1 || max_n < 1) return DSC_EINVAL;
    static_assert(sizeof(dsc_colsum_item) == sizeof(ColsumItem), "dsc_colsum_item layout");
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(colsum_grouped_kernel, dim3((max_n + 63) / 64, count), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const ColsumItem*>(items_dev));
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_gn_silu_bwd_f32(const float* z, int64_t ldz, const float* dy, int64_t ldy, const float* gamma,
                                   const float* beta, const float* scale_shift, int64_t ld_ss, int32_t ss_mode,
                                   float* dz, int64_t lddz, float* dgamma_p, float* dbeta_p, float* dbias_p,
                                   int64_t partial_stride, float* dss, int64_t ld_dss, int32_t scenes,
                                   int32_t tokens_per_scene, int32_t channels, float eps, dsc_stream_t stream) {
    if (!z || !dy || !gamma || !beta || !dz || !dgamma_p || !dbeta_p || !dbias_p) return DSC_EINVAL;
    if (scenes < 1 || tokens_per_scene < 1 || channels != 512) return DSC_ERANGE;
    if (ss_mode != DSC_SS_NONE && !scale_shift) return DSC_EINVAL;
    GnBwdArgs p{z, (long)ldz, dy, (long)ldy, gamma, beta, ss_mode != DSC_SS_NONE ? scale_shift : nullptr, (long)ld_ss,
                ss_mode, dz, (long)lddz, dgamma_p, dbeta_p, dbias_p, (long)partial_stride, dss, (long)ld_dss, tokens_per_scene,
                channels, eps};
    if (tokens_per_scene > 160) return DSC_ERANGE;
    const size_t lds = (size_t)2 * tokens_per_scene * 64 * sizeof(float);      // <= 80 KB
    DSC_CLEAR_STALE_ERROR();
    const bool v4 = ((ldz | ldy | lddz) & 3) == 0 && dsc_aligned16(z) && dsc_aligned16(dy) && dsc_aligned16(dz) &&
                    dsc_aligned16(gamma) && dsc_aligned16(beta) &&
                    (!p.ss || (dsc_aligned16(p.ss) && (ld_ss & 3) == 0)) &&
                    (!dss || (dsc_aligned16(dss) && (ld_dss & 3) == 0));
    if (lds > 48 * 1024) {
        static bool raised = false;
        if (!raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gn_silu_bwd_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e != hipSuccess) return (int)e;
            raised = true;
        }
    }
    hipStream_t hs = static_cast<hipStream_t>(stream);
    if (v4 && tokens_per_scene <= 32)
        hipLaunchKernelGGL(gn_silu_bwd_reg_kernel<2>, dim3(scenes * 8), dim3(256), 0, hs, p);
    else if (v4 && tokens_per_scene <= 80)
        hipLaunchKernelGGL(gn_silu_bwd_reg_kernel<5>, dim3(scenes * 8), dim3(256), 0, hs, p);
    else if (v4)
        hipLaunchKernelGGL(gn_silu_bwd_reg_kernel<10>, dim3(scenes * 8), dim3(256), 0, hs, p);
    else
        hipLaunchKernelGGL(gn_silu_bwd_kernel, dim3(scenes * 8), dim3(256), lds, static_cast<hipStream_t>(stream), p);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_weight_standardize_bwd_f32(const dsc_ws_bwd_item* items, int32_t count, float eps, dsc_stream_t stream) {
    if (!items || count < 1 || count > DSC_WS_MAX) return DSC_EINVAL;
    WsBwdBatch b;
    int maxrows = 0;
    for (int i = 0; i < count; ++i) {
        if (!items[i].w || !items[i].dw_std || !items[i].dw || items[i].rows < 1 || items[i].cols < 1) return DSC_EINVAL;
        if (items[i].cols > 2048) return DSC_ERANGE;
        b.it[i] = WsBwdItem{items[i].w, items[i].dw_std, items[i].dw, items[i].rows, items[i].cols};
        if (items[i].rows > maxrows) maxrows = items[i].rows;
    }
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(ws_bwd_kernel, dim3(maxrows, count), dim3(256), 0, static_cast<hipStream_t>(stream), b, eps);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_layernorm_bwd_f32(const float* x, int64_t ldx, const float* g, const float* dy, int64_t ldy, float* dx,
                                     int64_t lddx, const float* addend, int64_t ldadd, float* dg_partial, int32_t partial_rows,
                                     int32_t m, int32_t d, float eps, dsc_stream_t stream) {
    if (!x || !g || !dy || !dx || !dg_partial || m < 1 || partial_rows < 1) return DSC_EINVAL;
    if (d != 512) return DSC_ERANGE;
    if (!dsc_aligned16(x) || !dsc_aligned16(g) || !dsc_aligned16(dy) || !dsc_aligned16(dx) || (ldx & 3) || (ldy & 3) ||
        (lddx & 3) || (addend && (!dsc_aligned16(addend) || (ldadd & 3)))) return DSC_EALIGN;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(partial_rows), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, (long)ldx, g, dy, (long)ldy, dx, (long)lddx, addend, (long)ldadd, dg_partial, m, eps);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_linear_attention_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v,
                                            int64_t ldv, const float* dout, int64_t ldo, float* dq, int64_t lddq,
                                            float* dk, int64_t lddk, float* dv, int64_t lddv, int32_t scenes, int32_t nq,
                                            int32_t nk, float scale, dsc_stream_t stream) {
    if (!q || !k || !v || !dout || !dq || !dk || !dv || scenes < 1 || nq < 1 || nk < 1) return DSC_EINVAL;
    if (nq > 160 || nk > 160) return DSC_ERANGE;
    if ((ldq | ldk | ldv | ldo | lddq | lddk | lddv) & 3) return DSC_EALIGN;
    if (!dsc_aligned16(q) || !dsc_aligned16(k) || !dsc_aligned16(v) || !dsc_aligned16(dout) || !dsc_aligned16(dq) ||
        !dsc_aligned16(dk) || !dsc_aligned16(dv)) return DSC_EALIGN;
    const size_t lds = sizeof(float) * ((size_t)(3 * nk + 2 * nq) * LP + 2 * 32 * LP + 32);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_attention_bwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(linear_attention_bwd_kernel, dim3(scenes * DSC_HEADS), dim3(256), lds,
                       static_cast<hipStream_t>(stream), q, (long)ldq, k, (long)ldk, v, (long)ldv, dout, (long)ldo,
                       dq, (long)lddq, dk, (long)lddk, dv, (long)lddv, nq, nk, scale);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_attention_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                     const float* dout, int64_t ldo, float* dq, int64_t lddq, float* dk, int64_t lddk,
                                     float* dv, int64_t lddv, int32_t scenes, int32_t n, float scale, dsc_stream_t stream) {
    if (!q || !k || !v || !dout || !dq || !dk || !dv || scenes < 1 || n < 1) return DSC_EINVAL;
    if (n > 160) return DSC_ERANGE;
    if ((ldq | ldk | ldv | ldo | lddq | lddk | lddv) & 3) return DSC_EALIGN;
    if (!dsc_aligned16(q) || !dsc_aligned16(k) || !dsc_aligned16(v) || !dsc_aligned16(dout) || !dsc_aligned16(dq) ||
        !dsc_aligned16(dk) || !dsc_aligned16(dv)) return DSC_EALIGN;
    if (n <= 96) {
        const size_t ldc = sizeof(float) * ((size_t)4 * n * 36 + (size_t)2 * n * (n + 1));       // <= 128 KB
        static bool attr_c = false;
        if (!attr_c) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_cached_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            attr_c = true;
        }
        DSC_CLEAR_STALE_ERROR();
        hipLaunchKernelGGL(attention_bwd_cached_kernel, dim3(scenes * DSC_HEADS), dim3(dsc_attention_threads(n)), ldc,
                           static_cast<hipStream_t>(stream), q, (long)ldq, k, (long)ldk, v, (long)ldv, dout, (long)ldo, dq,
                           (long)lddq, dk, (long)lddk, dv, (long)lddv, n, scale);
        DSC_LAUNCH_CHECK();
        return 0;
    }
    const size_t lds = sizeof(float) * ((size_t)4 * n * HP + 3 * n);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(attention_bwd_kernel, dim3(scenes * DSC_HEADS), dim3(192), lds, static_cast<hipStream_t>(stream),
                       q, (long)ldq, k, (long)ldk, v, (long)ldv, dout, (long)ldo, dq, (long)lddq, dk, (long)lddk,
                       dv, (long)lddv, n, scale);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_activation_bwd_f32(const float* x, const float* dy, float* dx, int64_t count, int32_t act,
                                      dsc_stream_t stream) {
    if (!x || !dy || !dx || count < 1) return DSC_EINVAL;
    long blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, dy, dx,
                       (long)count, act);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_transpose_f32(const float* in, int64_t ldi, float* out, int64_t ldo, int32_t rows, int32_t cols,
                                 dsc_stream_t stream) {
    if (!in || !out || rows < 1 || cols < 1) return DSC_EINVAL;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0,
                       static_cast<hipStream_t>(stream), in, (long)ldi, out, (long)ldo, rows, cols);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_transpose_batched_f32(const dsc_ws_item* items, int32_t count, dsc_stream_t stream) {
    if (!items || count < 1 || count > DSC_WS_MAX) return DSC_EINVAL;
    TransposeBatch b;
    int maxr = 0, maxc = 0;
    for (int i = 0; i < count; ++i) {
        if (!items[i].w || !items[i].out || items[i].rows < 1 || items[i].cols < 1) return DSC_EINVAL;
        b.it[i] = items[i];
        if (items[i].rows > maxr) maxr = items[i].rows;
        if (items[i].cols > maxc) maxc = items[i].cols;
    }
    bool v4 = true;
    for (int i = 0; i < count; ++i)
        v4 = v4 && (items[i].rows % 4 == 0) && (items[i].cols % 4 == 0) && dsc_aligned16(items[i].w) && dsc_aligned16(items[i].out);
    DSC_CLEAR_STALE_ERROR();
    if (v4)
        hipLaunchKernelGGL(transpose_batched_v4_kernel, dim3((maxc + 63) / 64, (maxr + 63) / 64, count), dim3(256), 0,
                           static_cast<hipStream_t>(stream), b);
    else
        hipLaunchKernelGGL(transpose_batched_kernel, dim3((maxc + 31) / 32, (maxr + 31) / 32, count), dim3(256), 0,
                           static_cast<hipStream_t>(stream), b);
    DSC_LAUNCH_CHECK();
    return 0;
}

// =====================================================================================================
// Training loss of reference p_losses (diffusion_ddpm.py:556-652) -- forward AND d(loss)/d(denoise_out) in one kernel.
// One 256-thread block per scene:
//   separated MSE terms over the attribute slices, loss_weight[t] scaling, and (loss_iou) the pairwise axis-aligned
//   3-D IoU regulariser of loss.py:7-102 on the de-normalised, clamped x0 estimate, masked by predicted emptiness.
// parts[b] = {bbox, trans, size, angle, class, object, objfeat, liou, bbox_iou} of scene b; dout[b] = d losses_weight[b] / d out[b].
// =====================================================================================================
namespace {

struct LossArgs {
    const float* target; const float* out; const float* x_t; const int64_t* t;
    const float* loss_weight; const float* ca; const float* cb; const float* alphas_cumprod;
    float* losses; float* parts; float* dout;
    int n, c, tr, sz, bb, nc, no, nf;
    int separate, iou, mean_type;
    int T;                        // rows of the schedule tables: t[b] is held to [0, T)
    int arrange;                  // re-arrangement model: x = [translation | angle], losses = l_trans + l_angle (:558-571)
    float grad_scale;             // dout = grad_scale * d losses[b] / d out[b]   (1/B for loss = losses.mean())
    float c_lo[3], c_span[3], s_lo[3], s_span[3];
};

constexpr int LOSS_MAXN = 160;

__device__ __forceinline__ float block_sum_loss(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void ddpm_loss_kernel(const LossArgs p) {
    __shared__ float red[4];
    __shared__ float lo[LOSS_MAXN][3], hi[LOSS_MAXN][3], valid[LOSS_MAXN], giou[LOSS_MAXN][6];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = p.n, C = p.c;
    const long base = (long)b * N * C;
    const int64_t tv = dsc_checked_index(p.t[b], p.T, tid == 0);          // one block per scene: one count per out-of-range scene
    const float lw = p.loss_weight[tv];
    const int c_trans = p.tr, c_size = p.tr + p.sz, c_bbox = p.bb, c_class = p.bb + p.nc;
    const int c_obj0 = (p.no == 0) ? c_class - 1 : c_class, c_obj1 = c_class + p.no;
    // ---- squared-error sums per slice
    float s_tr = 0.f, s_sz = 0.f, s_an = 0.f, s_cl = 0.f, s_ob = 0.f, s_ft = 0.f, s_all = 0.f;
    for (int e = tid; e < N * C; e += 256) {
        const int ch = e % C;
        const float d = p.target[base + e] - p.out[base + e];
        const float q = d * d;
        s_all += q;
        if (ch < c_trans) s_tr += q;
        else if (ch < c_size) s_sz += q;
        else if (ch < c_bbox) s_an += q;
        else if (ch < c_class) s_cl += q;
        else if (ch >= c_obj1) s_ft += q;
        if (ch >= c_obj0 && ch < c_obj1) s_ob += q;
    }
    s_tr = block_sum_loss(s_tr, red); s_sz = block_sum_loss(s_sz, red); s_an = block_sum_loss(s_an, red);
    s_cl = block_sum_loss(s_cl, red); s_ob = block_sum_loss(s_ob, red); s_ft = block_sum_loss(s_ft, red);
    s_all = block_sum_loss(s_all, red);
    const float fn = (float)N;
    const float l_trans = s_tr / (fn * p.tr), l_size = p.sz > 0 ? s_sz / (fn * p.sz) : 0.f;
    const float l_angle = s_an / (fn * (p.bb - p.tr - p.sz));
    const float l_bbox = (s_tr + s_sz + s_an) / (fn * p.bb), l_class = p.nc > 0 ? s_cl / (fn * p.nc) : 0.f;
    const float l_obj = (c_obj1 > c_obj0 && !p.arrange) ? s_ob / (fn * (c_obj1 - c_obj0)) : 0.f;
    const float l_feat = p.nf > 0 ? s_ft / (fn * p.nf) : 0.f;
    float losses;
    if (p.arrange && p.separate) {
        losses = l_trans + l_angle;
    } else if (p.separate) {
        losses = l_bbox + l_class;
        if (p.no > 0) losses += l_obj;
        if (p.nf > 0) losses += l_feat;
    } else losses = s_all / (fn * C);
    float lossw = losses * lw;
    // ---- IoU regulariser
    float liou = 0.f, iou_avg = 0.f;
    const float A = (p.mean_type == DSC_MEAN_X0) ? 0.f : p.ca[tv];
    const float Bc = (p.mean_type == DSC_MEAN_X0) ? -1.f : p.cb[tv];     // x0 = A * x_t - Bc * out
    if (p.iou) {
        const int oc = (p.no > 0) ? c_class : c_class - 1;
        if (tid < N) {
            const long row = base + (long)tid * C;
            float xr[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const float raw = A * p.x_t[row + k] - Bc * p.out[row + k];
                xr[k] = fminf(fmaxf(raw, -1.0f), 1.0f);
            }
            const float ob = fminf(fmaxf(A * p.x_t[row + oc] - Bc * p.out[row + oc], -1.0f), 1.0f);
            valid[tid] = (p.no > 0) ? (ob >= 0.f ? 1.f : 0.f) : (ob <= 0.f ? 1.f : 0.f);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float ctr = (xr[k] + 1.0f) * 0.5f * p.c_span[k] + p.c_lo[k];
                const float siz = (xr[3 + k] + 1.0f) * 0.5f * p.s_span[k] + p.s_lo[k];
                lo[tid][k] = ctr - siz;
                hi[tid][k] = ctr + siz;
            }
        }
        __syncthreads();
        float S = 0.f, cnt = 0.f;
        float glo[3] = {0.f, 0.f, 0.f}, ghi[3] = {0.f, 0.f, 0.f};
        if (tid < N && valid[tid] > 0.f) {
            const int i = tid;
            float li[3], hi_i[3], ei[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { li[k] = lo[i][k]; hi_i[k] = hi[i][k]; ei[k] = hi_i[k] - li[k]; }
            const float vi = ei[0] * ei[1] * ei[2];
            for (int j = 0; j < N; ++j) {
                if (valid[j] <= 0.f) continue;
                cnt += 1.0f;
                float wh[3], ej[3];
                bool r_is_i[3], l_is_i[3], r_tie[3], l_tie[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float hj = hi[j][k], lj = lo[j][k];
                    ej[k] = hj - lj;
                    const float r = fminf(hi_i[k], hj), l = fmaxf(li[k], lj);
                    r_is_i[k] = hi_i[k] < hj; r_tie[k] = hi_i[k] == hj;
                    l_is_i[k] = li[k] > lj; l_tie[k] = li[k] == lj;
                    wh[k] = fmaxf(r - l, 0.f);
                }
                const float vj = ej[0] * ej[1] * ej[2];
                const float o = wh[0] * wh[1] * wh[2];
                const float uraw = vi + vj - o;
                const bool ucl = !(uraw > 1e-6f);                 // union clamped to eps
                const float u = ucl ? 1e-6f : uraw;
                S += o / u;
                if (j == i) {
                    // both arguments are box i: o = u = vi, the quotient is constant unless the union is clamped
                    if (ucl) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const float dv = ei[(k + 1) % 3] * ei[(k + 2) % 3] / u;
                            ghi[k] += dv; glo[k] -= dv;
                        }
                    }
                    continue;
                }
                // d iou / d box_i with box_j fixed; the (j, i) pair contributes the same amount (symmetry) -> factor 2
                const float dio = ucl ? 1.0f / u : (u + o) / (u * u);      // d iou / d o   (du/do = -1)
                const float diu = ucl ? 0.f : -o / (u * u);                // d iou / d vi  (through the union)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float others = wh[(k + 1) % 3] * wh[(k + 2) % 3];
                    const float pass = (fminf(hi_i[k], hi[j][k]) - fmaxf(li[k], lo[j][k]) >= 0.f) ? 1.f : 0.f;
                    const float dwh = dio * others * pass;                 // d iou / d wh_k
                    const float wr = r_tie[k] ? 0.5f : (r_is_i[k] ? 1.f : 0.f);
                    const float wl = l_tie[k] ? 0.5f : (l_is_i[k] ? 1.f : 0.f);
                    const float dvi = ei[(k + 1) % 3] * ei[(k + 2) % 3];   // d vi / d e_k
                    ghi[k] += 2.0f * (dwh * wr + diu * dvi);
                    glo[k] += 2.0f * (-dwh * wl - diu * dvi);
                }
            }
        }
        const float Ssum = block_sum_loss(S, red);
        const float csum = block_sum_loss(cnt, red) + 1e-6f;
        const float w = p.alphas_cumprod[tv];
        iou_avg = Ssum / csum;
        liou = w * 0.1f * Ssum / csum;
        lossw += liou;
        if (tid < N) {
            const float sc = (valid[tid] > 0.f) ? w * 0.1f / csum : 0.f;
            const long row = base + (long)tid * C;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float dctr = (glo[k] + ghi[k]) * sc, dsiz = (ghi[k] - glo[k]) * sc;
                const float raw_c = A * p.x_t[row + k] - Bc * p.out[row + k];
                const float raw_s = A * p.x_t[row + 3 + k] - Bc * p.out[row + 3 + k];
                const float pc = (raw_c >= -1.0f && raw_c <= 1.0f) ? 1.f : 0.f;
                const float ps = (raw_s >= -1.0f && raw_s <= 1.0f) ? 1.f : 0.f;
                giou[tid][k] = dctr * 0.5f * p.c_span[k] * pc * (-Bc);
                giou[tid][3 + k] = dsiz * 0.5f * p.s_span[k] * ps * (-Bc);
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        p.losses[b] = lossw;
        float* q = p.parts + (long)b * 9;
        q[0] = l_bbox; q[1] = l_trans; q[2] = l_size; q[3] = l_angle; q[4] = l_class; q[5] = l_obj; q[6] = l_feat;
        q[7] = liou; q[8] = iou_avg;
    }
    // ---- gradient of losses_weight[b] w.r.t. out[b]
    const float g_bbox = p.separate ? lw * 2.0f / (fn * p.bb) : lw * 2.0f / (fn * C);
    const float g_class = p.separate ? (p.nc > 0 ? lw * 2.0f / (fn * p.nc) : 0.f) : g_bbox;
    const float g_obj = p.separate ? (p.no > 0 ? lw * 2.0f / (fn * p.no) : 0.f) : g_bbox;
    const float g_feat = p.separate ? (p.nf > 0 ? lw * 2.0f / (fn * p.nf) : 0.f) : g_bbox;
    const float g_atr = lw * 2.0f / (fn * p.tr), g_aan = lw * 2.0f / (fn * (p.bb - p.tr - p.sz));   // arrange, separated
    for (int e = tid; e < N * C; e += 256) {
        const int ch = e % C, i = e / C;
        const float d = p.out[base + e] - p.target[base + e];
        float g;
        if (p.arrange && p.separate) g = (ch < c_trans) ? g_atr : g_aan;
        else g = (ch < c_bbox) ? g_bbox : (ch < c_class) ? g_class : (ch < c_obj1) ? g_obj : g_feat;
        g *= d;
        if (p.iou && ch < 6) g += giou[i][ch];
        p.dout[base + e] = g * p.grad_scale;
    }
}

}  // namespace

extern "C" int dsc_ddpm_loss_f32(const float* target, const float* out, const float* x_t, const int64_t* t,
                                 const float* loss_weight, const float* ca, const float* cb, const float* alphas_cumprod,
                                 const float* bounds /* host: c_lo[3] c_hi[3] s_lo[3] s_hi[3], may be NULL */,
                                 float* losses, float* parts, float* dout, int32_t b, int32_t n, int32_t c,
                                 int32_t translation_dim, int32_t size_dim, int32_t bbox_dim, int32_t class_dim,
                                 int32_t objectness_dim, int32_t objfeat_dim, int32_t loss_separate, int32_t loss_iou,
                                 int32_t mean_type, float grad_scale, int32_t num_timesteps, dsc_stream_t stream) {
    if (!target || !out || !x_t || !t || !loss_weight || !losses || !parts || !dout || b < 1 || n < 1 || num_timesteps < 1) return DSC_EINVAL;
    if (n > LOSS_MAXN) return DSC_ERANGE;
    if (c != bbox_dim + class_dim + objectness_dim + objfeat_dim) return DSC_EINVAL;
    // the re-arrangement model diffuses [translation | angle] only (diffusion_ddpm.py:558-571): no size / class /
    // objectness / shape-code channels, no IoU term
    const bool arrange = size_dim == 0 && class_dim == 0 && objectness_dim == 0 && objfeat_dim == 0;
    if (translation_dim != 3 || (size_dim != 3 && !arrange)) return DSC_ERANGE;
    if (arrange && (loss_iou || bbox_dim <= translation_dim)) return DSC_EINVAL;
    if (loss_iou && (!bounds || !alphas_cumprod)) return DSC_EINVAL;
    if (mean_type != DSC_MEAN_X0 && (!ca || !cb)) return DSC_EINVAL;
    LossArgs p{};
    p.target = target; p.out = out; p.x_t = x_t; p.t = t; p.loss_weight = loss_weight; p.ca = ca; p.cb = cb;
    p.alphas_cumprod = alphas_cumprod; p.losses = losses; p.parts = parts; p.dout = dout;
    p.n = n; p.c = c; p.tr = translation_dim; p.sz = size_dim; p.bb = bbox_dim; p.nc = class_dim; p.no = objectness_dim;
    p.nf = objfeat_dim; p.separate = loss_separate; p.iou = loss_iou; p.mean_type = mean_type;
    p.arrange = arrange ? 1 : 0;
    p.T = num_timesteps;
    p.grad_scale = grad_scale;
    if (bounds)
        for (int k = 0; k < 3; ++k) {
            p.c_lo[k] = bounds[k]; p.c_span[k] = bounds[3 + k] - bounds[k];
            p.s_lo[k] = bounds[6 + k]; p.s_span[k] = bounds[9 + k] - bounds[6 + k];
        }
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(ddpm_loss_kernel, dim3(b), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    DSC_LAUNCH_CHECK();
    return 0;
}

unsigned dsc_bad_index_train(bool reset) { return dsc_read_bad_index_count(reset); }

// ------------------------------------------------------------------------------------------------------
// Strided 2-D helpers of the static training plan (train_plan.py): gradient accumulation of multi-consumer activations
// (skip connections, residuals) and staging of un-aligned column slices.  16-byte path when everything is aligned.
// ------------------------------------------------------------------------------------------------------
namespace {

template <bool ADD>
__global__ __launch_bounds__(256) void copy2d_kernel(float* __restrict__ dst, long ldd, const float* __restrict__ src, long lds,
                                                     int rows, int cols, int vec) {
    const long total = vec ? (long)rows * (cols >> 2) : (long)rows * cols;
    const long stride = (long)gridDim.x * blockDim.x;
    if (vec) {
        const int c4n = cols >> 2;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
            const long r = i / c4n;
            const int c = (int)(i - r * c4n) * 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(src + r * lds + c);
            if (ADD) v += *reinterpret_cast<const f32x4*>(dst + r * ldd + c);
            *reinterpret_cast<f32x4*>(dst + r * ldd + c) = v;
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
            const long r = i / cols;
            const int c = (int)(i - r * cols);
            float v = src[r * lds + c];
            if (ADD) v += dst[r * ldd + c];
            dst[r * ldd + c] = v;
        }
    }
}

template <bool ADD>
int launch_copy2d(float* dst, int64_t ldd, const float* src, int64_t lds, int32_t rows, int32_t cols, dsc_stream_t stream) {
    if (!dst || !src || rows < 1 || cols < 1 || ldd < cols || lds < cols) return DSC_EINVAL;
    const int vec = ((cols | ldd | lds) & 3) == 0 && dsc_aligned16(dst) && dsc_aligned16(src);
    const long total = vec ? (long)rows * (cols >> 2) : (long)rows * cols;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(copy2d_kernel<ADD>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), dst,
                       (long)ldd, src, (long)lds, rows, cols, vec);
    DSC_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int dsc_copy2d_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int32_t rows, int32_t cols,
                              dsc_stream_t stream) {
    return launch_copy2d<false>(dst, ldd, src, lds, rows, cols, stream);
}

extern "C" int dsc_add2d_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int32_t rows, int32_t cols,
                             dsc_stream_t stream) {
    return launch_copy2d<true>(dst, ldd, src, lds, rows, cols, stream);
}

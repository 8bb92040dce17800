#pragma once
// Weight-gradient GEMM on the bf16 matrix cores with f32 accuracy (the "split-bf16" arithmetic of gemm_split.hip: both operands split
// exactly into three bf16 pieces, six products, f32 accumulation), grouped over many layers like gemm_tn_grouped_kernel.
//
//   dw[n][k] = sum_m dy[m][n] * [a1 | a2][m][k]              dy, a1, a2: f32 activations in HBM; dw: f32 (slice of the flat gradient G)
//
// The reduction runs over TOKENS, the slow dimension of both operands: an MFMA fragment needs 8 consecutive tokens of one channel.
// So the split happens once per block at staging: a thread loads 8 consecutive tokens of ONE channel (8 dword loads, coalesced
// across the lanes' consecutive channels), splits them and writes three 16-byte bf16x8 pieces into channel-major LDS planes
// [3][384 channels][32 tokens].  Fragments are then one 16-byte read per plane for both operands.
//
// Block: 256 dy-channels (n) x 128 x-channels (k) of the output, one slice of the token range (blockIdx.y; slices > 1 write slabs
// that reduce_grouped_kernel sums in a fixed order), 8 waves as 4 (n) x 2 (k), wave tile 64 x 64 = 4 x 4 MFMA blocks of 16 x 16,
// 32 tokens per step, two LDS stages of 72 KiB.  The MFMA computes dw^T blocks (x as the row operand) so that a lane holds 4
// consecutive k of one n: 16-byte stores.  Channels past n / K and tokens past m need no masking code: buffer loads past the end of
// an operand return zeros, and whatever a channel past the edge reads only reaches output rows / columns that are never stored.
// The bias gradient (column sums of dy) rides in the staging registers of the k-tile-0 blocks.
#include "dsc_common.h"

namespace dsc_tn_split {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int BN = 256, BKO = 128, COLS = BN + BKO, BMS = 32, NW = 8, T = NW * 64;
constexpr int PLANE = COLS * BMS * 2;                 // 24576 B: one bf16 plane, channel-major (64 B per channel)
constexpr int STAGE = 3 * PLANE;                      // 73728 B
constexpr int NIT = COLS * 4 / T;                     // 3 (channel, token-octet) items per thread and step
constexpr int SMEM = 2 * STAGE + 4 * BN * 4;          // + [4 token octets][256 channels] f32 for the bias gradient

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& p1, bf16x8& p2, bf16x8& p3) {
    u32x4 a, b, c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x0 = x[2 * q], x1 = x[2 * q + 1];
        const unsigned u1 = cvt_pk_bf16(x0, x1);
        const float r0 = x0 - bf_lo(u1), r1 = x1 - bf_hi(u1);
        const unsigned u2 = cvt_pk_bf16(r0, r1);
        a[q] = u1;
        b[q] = u2;
        c[q] = cvt_pk_bf16(r0 - bf_lo(u2), r1 - bf_hi(u2));
    }
    p1 = __builtin_bit_cast(bf16x8, a);
    p2 = __builtin_bit_cast(bf16x8, b);
    p3 = __builtin_bit_cast(bf16x8, c);
}

struct Prob {
    const float* a1; long lda1; int k1;
    const float* a2; long lda2; int k2;
    const float* dy; long ldd;
    float* out; long ldo; float* bias_out;
    int m, n, kvalid;
    long chunk, slab, bias_slab;
};

// it: 128-wide k tile, jt: 256-wide n tile, split: token slice.
// PROBE (tools/tn_probe.hip only; 0 = product): attribution variants that REMOVE one ingredient of the step (results are garbage) --
// 1: no operand split (raw bits written as the three pieces), 2: no global loads, 3: no fragment reads (registers reused),
// 4: 2 + 1 + no plane writes (MFMAs, fragment reads and the barrier only), 5: MFMAs and the barrier only.
template <int PROBE = 0>
__device__ __forceinline__ void tn_split_block(const Prob& p, const int it, const int jt, const int split, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wn = wave_u & 3, wk = wave_u >> 2;
    const int n0 = jt * BN, k0 = it * BKO;
    const long m_begin = (long)split * p.chunk;
    const long m_end = (m_begin + p.chunk < p.m) ? m_begin + p.chunk : p.m;
    const int steps = m_end > m_begin ? (int)((m_end - m_begin + BMS - 1) / BMS) : 0;
    const int g = lane >> 4, l15 = lane & 15;

    // the x operand of this k tile: one of the two K segments (a K tile never straddles them: k1 % 128 == 0 when k2 > 0)
    const bool seg1 = k0 < p.k1;
    const float* const xbase = seg1 ? p.a1 + k0 : p.a2 + (k0 - p.k1);
    const long ldx = seg1 ? p.lda1 : p.lda2;
    const int xcols = seg1 ? p.k1 - k0 : p.k1 + p.k2 - k0;          // valid columns from xbase on
    const float* const dbase = p.dy + n0;
    const int dcols = p.n - n0;
#if defined(__HIP_DEVICE_COMPILE__)
    // num_records = bytes up to the end of the last valid row: token rows >= m read zeros (only the last slice has a ragged tail:
    // chunk is a multiple of 32)
    long xrec = ((long)(p.m - 1) * ldx + xcols) * 4, drec = ((long)(p.m - 1) * p.ldd + dcols) * 4;
    if (xrec > 0x7fffffffL) xrec = 0x7fffffffL;
    if (drec > 0x7fffffffL) drec = 0x7fffffffL;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xbase), 0, (int)xrec, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dbase), 0, (int)drec, 0x00020000);
#endif

    // staging items: t = u*512 + tid -> channel c = t % 384 (0..255: dy channel n0+c, 256..383: x channel k0+c-256), token octet t / 384
    int ivoff[NIT], ildso[NIT], bslot[NIT];
    bool isdy[NIT];
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int t = u * T + tid, c = t % COLS, og = t / COLS;
        isdy[u] = __builtin_amdgcn_readfirstlane((u * T + wave_u * 64) % COLS) < BN;       // wave-uniform: 384 = 6 x 64
        // channels past the edge of the operand read channel 0 (valid memory; their products are never stored)
        const int cc = isdy[u] ? (c < dcols ? c : 0) : (c - BN < xcols ? c - BN : 0);
        ivoff[u] = (int)((cc + 8L * og * (isdy[u] ? p.ldd : ldx)) * 4);
        ildso[u] = c * 64 + ((og ^ ((c >> 1) & 3)) << 4);
        bslot[u] = og * BN + c;
    }
    const bool do_bias = p.bias_out != nullptr && it == 0;
    float bsum[NIT] = {0.f, 0.f, 0.f};
    float ld[NIT][8];
    auto load_item = [&](int u, int step) {
        const long mb = m_begin + (long)step * BMS;
        if constexpr (PROBE == 2 || PROBE >= 4) return;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int e = 0; e < 8; ++e)
            ld[u][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(isdy[u] ? rdy : rx, ivoff[u],
                                                                                       (int)((mb + e) * (isdy[u] ? p.ldd : ldx) * 4), 0));
#else
        (void)mb;
#endif
    };
    auto store_item = [&](int u, char* stage, bool count) {
        if (do_bias && isdy[u] && count) bsum[u] += ((ld[u][0] + ld[u][1]) + (ld[u][2] + ld[u][3])) + ((ld[u][4] + ld[u][5]) + (ld[u][6] + ld[u][7]));
        bf16x8 a, b, c;
        if constexpr (PROBE >= 4) return;
        if constexpr (PROBE == 1) {
            u32x4 lo = {__builtin_bit_cast(unsigned, ld[u][0]), __builtin_bit_cast(unsigned, ld[u][1]), __builtin_bit_cast(unsigned, ld[u][2]),
                        __builtin_bit_cast(unsigned, ld[u][3])};
            u32x4 hi = {__builtin_bit_cast(unsigned, ld[u][4]), __builtin_bit_cast(unsigned, ld[u][5]), __builtin_bit_cast(unsigned, ld[u][6]),
                        __builtin_bit_cast(unsigned, ld[u][7])};
            a = __builtin_bit_cast(bf16x8, lo); b = __builtin_bit_cast(bf16x8, hi); c = a;
        } else {
            split8(ld[u], a, b, c);
        }
        *reinterpret_cast<bf16x8*>(stage + ildso[u]) = a;
        *reinterpret_cast<bf16x8*>(stage + PLANE + ildso[u]) = b;
        *reinterpret_cast<bf16x8*>(stage + 2 * PLANE + ildso[u]) = c;
    };

    // fragment offsets: x (row operand): channel 256 + wk*64 + kb*16 + l15; dy (column operand): channel wn*64 + nb*16 + l15
    int xoff[4], doff[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int cx = BN + wk * 64 + b * 16 + l15, cd = wn * 64 + b * 16 + l15;
        xoff[b] = cx * 64 + ((g ^ ((cx >> 1) & 3)) << 4);
        doff[b] = cd * 64 + ((g ^ ((cd >> 1) & 3)) << 4);
    }
    f32x4 acc[4][4];                                  // [kb][nb]: lane = (n = l15 of block nb, k = 4g..4g+3 of block kb)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int NMMA = 24;                          // MFMAs per dy block (4 x blocks x 6 products)
    bf16x8 pf_x[4][3], pf_d[3];                       // (probes 3 / 5 only: dead code in the product)
    if constexpr (PROBE != 0) {
#pragma unroll
        for (int u = 0; u < NIT; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) ld[u][e] = (float)(tid + e) * 1e-3f;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            pf_d[pl] = *reinterpret_cast<const bf16x8*>(smem + pl * PLANE + doff[0]);
#pragma unroll
            for (int b = 0; b < 4; ++b) pf_x[b][pl] = *reinterpret_cast<const bf16x8*>(smem + pl * PLANE + xoff[b]);
        }
    }
    if (steps > 0) {
#pragma unroll
        for (int u = 0; u < NIT; ++u) load_item(u, 0);
#pragma unroll
        for (int u = 0; u < NIT; ++u) store_item(u, smem, true);
#pragma unroll
        for (int u = 0; u < NIT; ++u) load_item(u, steps > 1 ? 1 : 0);       // step 1: split under step 0
    }
    for (int s = 0; s < steps; ++s) {
        __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): my plane writes of step s are done
        __syncthreads();                              // everyone's are; nobody reads the other stage any more
        char* cur = smem + (s & 1) * STAGE;
        char* nxt = smem + ((s + 1) & 1) * STAGE;
        // item u of step s + 1 (requested a whole step ago, right after the same registers were consumed) is split under dY block
        // u + 1 of this step, and its registers immediately take the request for step s + 2: one step of load lead without a second
        // register set.  Past the end the rows are duplicates: written to a stage nobody reads, never counted into the bias sums.
        const bool more = s + 1 < steps;
        const int s2 = s + 2 < steps ? s + 2 : steps - 1;
        bf16x8 xf[4][3], df[2][3];
        constexpr bool NOFRAG = PROBE == 3 || PROBE == 5;
        if constexpr (NOFRAG) {               // probe: the fragments of step 0 stay in registers, no LDS read traffic in the loop
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) { df[0][pl] = pf_d[pl]; df[1][pl] = pf_d[pl]; }
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) xf[b][pl] = pf_x[b][pl];
        } else {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) df[0][pl] = *reinterpret_cast<const bf16x8*>(cur + pl * PLANE + doff[0]);
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) xf[b][pl] = *reinterpret_cast<const bf16x8*>(cur + pl * PLANE + xoff[b]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            if (nb + 1 < 4 && !NOFRAG) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) df[(nb + 1) & 1][pl] = *reinterpret_cast<const bf16x8*>(cur + pl * PLANE + doff[nb + 1]);
            }
            if (nb >= 1) {
                // the 8 loads of item nb - 1 are the oldest in flight; the two younger items (16 loads) may keep flying
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0x0f70 | 0 | (1 << 14));                    // vmcnt(16)
                __builtin_amdgcn_sched_barrier(0);
                store_item(nb - 1, nxt, more);
                load_item(nb - 1, s2);
            }
            const bf16x8 (&d)[3] = df[nb & 1];
            // product-major over the 4 x blocks: an accumulator comes round every 4th MFMA; small terms first
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[kb][2], d[0], acc[kb][nb], 0, 0, 0);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[kb][0], d[2], acc[kb][nb], 0, 0, 0);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[kb][1], d[1], acc[kb][nb], 0, 0, 0);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[kb][1], d[0], acc[kb][nb], 0, 0, 0);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[kb][0], d[1], acc[kb][nb], 0, 0, 0);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[kb][0], d[0], acc[kb][nb], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (nb + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            if (nb >= 1) {
#pragma unroll
                for (int q = 0; q < NMMA - 2; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, (44 + NMMA - 3) / (NMMA - 2), 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);

    float* out = p.out + (long)split * p.slab;
    if (do_bias) {
        // every dy channel has four items (token octets) in four different threads: partial sums through LDS, fixed order
        float* bs = reinterpret_cast<float*>(smem + 2 * STAGE);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NIT; ++u)
            if (isdy[u]) bs[bslot[u]] = bsum[u];
        __syncthreads();
        if (tid < BN && n0 + tid < p.n)
            p.bias_out[(long)split * p.bias_slab + n0 + tid] = (bs[tid] + bs[BN + tid]) + (bs[2 * BN + tid] + bs[3 * BN + tid]);
    }
    const bool vec = (p.ldo & 3) == 0 && (p.slab & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int j = n0 + wn * 64 + nb * 16 + l15;               // output row (n)
        if (j >= p.n) continue;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const int i = k0 + wk * 64 + kb * 16 + 4 * g;         // output column (k), 4 consecutive
            if (vec && i + 3 < p.kvalid) {
                *reinterpret_cast<f32x4*>(out + (long)j * p.ldo + i) = acc[kb][nb];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (i + e < p.kvalid) out[(long)j * p.ldo + i + e] = acc[kb][nb][e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the same block with PRODUCER and CONSUMER waves.  An MFMA blocks its wave: in tn_split_block every wave interleaves its
// 96 MFMAs per step with its own share of the staging (24 loads, 132 VALU of operand split, 9 LDS writes), and the round-4 probes
// showed those costs ADD to the MFMA time (5156 cycles per step against an MFMA floor of 3072) however they are slotted -- the two
// waves of a SIMD walk in phase.  Here the block has 12 waves: waves 0-7 are the round-4 consumers (fragment reads + MFMAs only: same
// wave tiles, same fragment offsets, same MFMA order, hence bit-identical dW), waves 8-11 do ALL the staging of the block (the
// same (channel, token octet) items, six per lane and step instead of three, the same LDS image, the same per-item bias partial sums
// in the same LDS slots, hence bit-identical bias gradients) one step ahead, and the only thing the two kinds share is the one barrier
// per step.  A SIMD holds two consumers and one producer: the producer's VALU / VMEM / LDS-write instructions issue in the cycles
// the matrix pipe is busy with a consumer's MFMA.  168 VGPRs per wave (three waves per SIMD).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int NPW = 4, TP = NPW * 64;                 // producer waves / threads
constexpr int PAIRS = COLS / 2, NITX = PAIRS * 4 / TP;// 192 channel pairs x 4 token octets = 3 items per producer lane and step (8-byte form)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// NCW = consumer waves: 8 (wave tile 64 n x 64 k, the round-4 tiles; 768 threads) or 4 (wave tile 64 n x 128 k: the dY fragments of
// all four n blocks stay in registers and the x fragments stream per k block -- 36 instead of 48 fragment reads per 192 MFMAs, i.e.
// 144 instead of 196 KB of LDS reads per step and CU; 512 threads).  Every accumulator sees the same six products in the same order.
template <int NCW>
__device__ __forceinline__ void tn_split_block_ws(const Prob& p, const int it, const int jt, const int split, char* smem) {
    static_assert(NCW == 8 || NCW == 4, "8 or 4 consumer waves");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = jt * BN, k0 = it * BKO;
    const long m_begin = (long)split * p.chunk;
    const long m_end = (m_begin + p.chunk < p.m) ? m_begin + p.chunk : p.m;
    const int steps = m_end > m_begin ? (int)((m_end - m_begin + BMS - 1) / BMS) : 0;
    const bool do_bias = p.bias_out != nullptr && it == 0;
    float* const bs = reinterpret_cast<float*>(smem + 2 * STAGE);

    if (wave_u >= NCW) {
        // ------------------------------------------------------------------------------------------------ producers
#ifdef DSC_TN_PRIO
        __builtin_amdgcn_s_setprio(DSC_TN_PRIO);
#endif
        const int pt = tid - NCW * 64;                    // 0..255
        const int pw = wave_u - NCW;
        const bool seg1 = k0 < p.k1;
        const float* const xbase = seg1 ? p.a1 + k0 : p.a2 + (k0 - p.k1);
        const long ldx = seg1 ? p.lda1 : p.lda2;
        const int xcols = seg1 ? p.k1 - k0 : p.k1 + p.k2 - k0;
        const float* const dbase = p.dy + n0;
        const int dcols = p.n - n0;
#if defined(__HIP_DEVICE_COMPILE__)
        long xrec = ((long)(p.m - 1) * ldx + xcols) * 4, drec = ((long)(p.m - 1) * p.ldd + dcols) * 4;
        if (xrec > 0x7fffffffL) xrec = 0x7fffffffL;
        if (drec > 0x7fffffffL) drec = 0x7fffffffL;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xbase), 0, (int)xrec, 0x00020000);
        const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dbase), 0, (int)drec, 0x00020000);
#endif
        {
            // 8-byte loads, TWO steps of lead.  (The host guarantees 16-byte aligned operands, leading dimensions and channel counts that are
            // multiples of 4: train_plan.HipBackend.gemm_tn_grouped.)  An item is a channel PAIR x a token octet -- 8 loads of 8 bytes, both
            // channels of a token at once: 24 loads per lane and step, so that two steps in flight fit the 6-bit vmcnt -- three items per
            // lane and step, two register sets alternating by step.  (Probes, profiles/r05_tn_producer_consumer.txt: producers idle 3830 us
            // for the launch = the MFMA floor; split + plane writes 4459; + L1-hit loads 5334; + real loads 6206.)
            int ivoff[NITX], ildso[NITX], bslot[NITX];
            bool isdy[NITX];
#pragma unroll
            for (int u = 0; u < NITX; ++u) {
                const int t = u * TP + pt, pr = t % PAIRS, og = t / PAIRS;
                isdy[u] = __builtin_amdgcn_readfirstlane((u * TP + pw * 64) % PAIRS) < BN / 2;     // wave-uniform: 192 = 3 x 64
                const int c = isdy[u] ? 2 * pr : 2 * pr - BN;                                       // channel inside its operand
                const int cc = isdy[u] ? (c < dcols ? c : 0) : (c < xcols ? c : 0);                 // past the edge: channel 0 (never stored)
                ivoff[u] = (int)((cc + 8L * og * (isdy[u] ? p.ldd : ldx)) * 4);
                ildso[u] = 2 * pr * 64 + ((og ^ (pr & 3)) << 4);                                    // LDS channel 2 pr; 2 pr + 1 sits 64 bytes on
                bslot[u] = og * BN + 2 * pr;
            }
            float bsum[NITX][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
            float ld[2][NITX][2][8];                          // [register set][item][channel of the pair][token]
            auto load_item = [&](auto setc, int u, int step) {
                constexpr int S = decltype(setc)::value;
#if defined(__HIP_DEVICE_COMPILE__)
                const long mb = m_begin + (long)step * BMS;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(isdy[u] ? rdy : rx, ivoff[u], (int)((mb + e) * (isdy[u] ? p.ldd : ldx) * 4), 0);
                    const unsigned lo = v[0], hi = v[1];      // (bit-casting the element expression itself reads element 0 twice: clang)
                    ld[S][u][0][e] = __builtin_bit_cast(float, lo);
                    ld[S][u][1][e] = __builtin_bit_cast(float, hi);
                }
#else
                (void)u; (void)step;
#endif
            };
            auto store_item = [&](auto setc, int u, char* stage, bool count) {
                constexpr int S = decltype(setc)::value;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float (&x)[8] = ld[S][u][h];
                    if (do_bias && isdy[u] && count) bsum[u][h] += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
                    bf16x8 a, b, c;
                    split8(x, a, b, c);
                    *reinterpret_cast<bf16x8*>(stage + ildso[u] + 64 * h) = a;
                    *reinterpret_cast<bf16x8*>(stage + PLANE + ildso[u] + 64 * h) = b;
                    *reinterpret_cast<bf16x8*>(stage + 2 * PLANE + ildso[u] + 64 * h) = c;
                }
            };
            using S0 = std::integral_constant<int, 0>;
            using S1 = std::integral_constant<int, 1>;
            auto clampi = [&](int st) { return st < steps ? st : steps - 1; };
            if (steps > 0) {
#pragma unroll
                for (int u = 0; u < NITX; ++u) load_item(S0{}, u, 0);
#pragma unroll
                for (int u = 0; u < NITX; ++u) store_item(S0{}, u, smem, true);
#pragma unroll
                for (int u = 0; u < NITX; ++u) load_item(S1{}, u, clampi(1));
#pragma unroll
                for (int u = 0; u < NITX; ++u) load_item(S0{}, u, clampi(2));
            }
            // step s: the data of step s + 1 (set (s + 1) & 1, requested two steps ago) goes to the other stage; its registers take step s + 3
            auto pstep = [&](auto setc, int s) {
                __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): my plane writes of step s are done
                __syncthreads();                          // the consumers have left the other stage
                char* nxt = smem + ((s + 1) & 1) * STAGE;
                const bool more = s + 1 < steps;
                const int s3 = clampi(s + 3);
#pragma unroll
                for (int u = 0; u < NITX; ++u) {
                    // younger than item u of this set: the rest of the set, the whole other set, and what was re-requested so far: 40 loads
                    __builtin_amdgcn_s_waitcnt(0x0f70 | (40 & 15) | ((40 >> 4) << 14));
                    store_item(setc, u, nxt, more);
#ifdef DSC_TN_WS_SAMEROWS
                    load_item(setc, u, 0);            // tools only: same instruction stream, every load an L1 hit (results are garbage)
#else
                    load_item(setc, u, s3);
#endif
                }
            };
            for (int s = 0; s < steps; s += 2) {
                pstep(S1{}, s);
                if (s + 1 < steps) pstep(S0{}, s + 1);
            }
            __builtin_amdgcn_s_waitcnt(0x0070);
            if (do_bias) {
                __syncthreads();
#pragma unroll
                for (int u = 0; u < NITX; ++u)
                    if (isdy[u]) { bs[bslot[u]] = bsum[u][0]; bs[bslot[u] + 1] = bsum[u][1]; }
                __syncthreads();
            }
            return;
        }
    }

    // ---------------------------------------------------------------------------------------------------- consumers
    constexpr int KBW = 32 / NCW;                          // 16-channel k blocks per wave: 4 (64 k) or 8 (128 k)
    const int wn = wave_u & 3, wk = wave_u >> 2;           // NCW == 4: wk == 0
    const int g = lane >> 4, l15 = lane & 15;
    int xoff[KBW], doff[4];
#pragma unroll
    for (int b = 0; b < KBW; ++b) {
        const int cx = BN + wk * 64 + b * 16 + l15;
        xoff[b] = cx * 64 + ((g ^ ((cx >> 1) & 3)) << 4);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int cd = wn * 64 + b * 16 + l15;
        doff[b] = cd * 64 + ((g ^ ((cd >> 1) & 3)) << 4);
    }
    f32x4 acc[KBW][4];
#pragma unroll
    for (int a = 0; a < KBW; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < steps; ++s) {
        __syncthreads();                                  // stage s & 1 is complete
        char* cur = smem + (s & 1) * STAGE;
        if constexpr (NCW == 8) {
            bf16x8 xf[4][3], df[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) df[0][pl] = *reinterpret_cast<const bf16x8*>(cur + pl * PLANE + doff[0]);
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) xf[b][pl] = *reinterpret_cast<const bf16x8*>(cur + pl * PLANE + xoff[b]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                if (nb + 1 < 4) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) df[(nb + 1) & 1][pl] = *reinterpret_cast<const bf16x8*>(cur + pl * PLANE + doff[nb + 1]);
                }
                const bf16x8 (&d)[3] = df[nb & 1];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[kb][2], d[0], acc[kb][nb], 0, 0, 0);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[kb][0], d[2], acc[kb][nb], 0, 0, 0);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[kb][1], d[1], acc[kb][nb], 0, 0, 0);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[kb][1], d[0], acc[kb][nb], 0, 0, 0);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[kb][0], d[1], acc[kb][nb], 0, 0, 0);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[kb][0], d[0], acc[kb][nb], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (nb + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 23, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            bf16x8 df[4][3], xf[2][3];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) df[nb][pl] = *reinterpret_cast<const bf16x8*>(cur + pl * PLANE + doff[nb]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) xf[0][pl] = *reinterpret_cast<const bf16x8*>(cur + pl * PLANE + xoff[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < KBW; ++kb) {
                if (kb + 1 < KBW) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) xf[(kb + 1) & 1][pl] = *reinterpret_cast<const bf16x8*>(cur + pl * PLANE + xoff[kb + 1]);
                }
                const bf16x8 (&x)[3] = xf[kb & 1];
                // the same six products in the same order per accumulator as the 8-wave form; here the four n blocks rotate
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[2], df[nb][0], acc[kb][nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[0], df[nb][2], acc[kb][nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[1], df[nb][1], acc[kb][nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[1], df[nb][0], acc[kb][nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[0], df[nb][1], acc[kb][nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[0], df[nb][0], acc[kb][nb], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (kb + 1 < KBW) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 23, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    float* out = p.out + (long)split * p.slab;
    if (do_bias) {
        __syncthreads();                                  // (the producers write their partial sums between these two)
        __syncthreads();
        if (tid < BN && n0 + tid < p.n)
            p.bias_out[(long)split * p.bias_slab + n0 + tid] = (bs[tid] + bs[BN + tid]) + (bs[2 * BN + tid] + bs[3 * BN + tid]);
    }
    const bool vec = (p.ldo & 3) == 0 && (p.slab & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int j = n0 + wn * 64 + nb * 16 + l15;
        if (j >= p.n) continue;
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb) {
            const int i = k0 + wk * 64 + kb * 16 + 4 * g;
            if (vec && i + 3 < p.kvalid) {
                *reinterpret_cast<f32x4*>(out + (long)j * p.ldo + i) = acc[kb][nb];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (i + e < p.kvalid) out[(long)j * p.ldo + i + e] = acc[kb][nb][e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the 256 x 256 block tile.  Rounds 4 / 5 showed the token step of the 256 x 128 tile bound by the BYTES it moves per MFMA --
// through the vector-memory path, the LDS writes and the fragment reads -- whoever issues them (producer waves at the MFMA floor when
// nothing is staged; every ingredient of the staging costs its full time).  This body halves them: four waves, ONE per SIMD with the
// 512-register budget, wave tile 128 n x 128 k = 8 x 8 MFMA blocks (256 accumulator registers); per step and block 64 KiB of operand
// rows for 1536 MFMAs (48 KiB for 768 before), 48 fragment reads per 384 MFMAs and wave (36 per 192).  A wave holds the 24 dY
// fragments of the step and streams the x fragments per k block; the staging (the same (channel pair, token octet) items, the same
// exact split, the same LDS image, the same per-item bias partial sums) rides between the wave's own MFMAs, placed by hand (one MFMA,
// one filler, one scheduling barrier -- as in gemm_split_wave.h), one step ahead.
// LDS: two stages of the first two piece planes (4 x 32 KiB) and ONE buffer for the third-piece plane (32 KiB) = all 160 KiB.  The
// third pieces only enter the first two products of a block (x3 d1, x1 d3): by the last k block of a step every wave has read the last
// x3 fragment and holds its d3 fragments in registers, so a barrier there frees the plane and the next step's third pieces -- kept in
// 32 registers until then -- are written under the last k block's MFMAs.  Two barriers per step, no exposed write phase.
// Every accumulator sees the same six products per token step in the same order as in the other two bodies: bit-identical dW and bias
// gradients (tests/test_gpu_train.py holds the three to each other).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int BBN = 256, BBK = 256, BCOLS = BBN + BBK, BT = 256;
constexpr int BPLANE = BCOLS * BMS * 2;               // 32768 B: one bf16 plane of the 512 staged channels
constexpr int BSMEM = 5 * BPLANE;                     // 163840 B = the whole LDS
constexpr int BITEMS = (BCOLS / 2) * 4 / BT;          // 4 (channel pair, token octet) items per lane and step: lane = pair, item = octet

__device__ __forceinline__ void tn_split_block_big(const Prob& p, const int it, const int jt, const int split, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave_u & 1, wk = wave_u >> 1;
    const int n0 = jt * BBN, k0 = it * BBK;
    const long m_begin = (long)split * p.chunk;
    const long m_end = (m_begin + p.chunk < p.m) ? m_begin + p.chunk : p.m;
    const int steps = m_end > m_begin ? (int)((m_end - m_begin + BMS - 1) / BMS) : 0;
    const bool do_bias = p.bias_out != nullptr && it == 0;
    const int g = lane >> 4, l15 = lane & 15;

    // operands of this tile (a k tile never straddles the two K segments: k1 % 256 == 0 when k2 > 0, checked by the host)
    const bool seg1 = k0 < p.k1;
    const float* const xbase = seg1 ? p.a1 + k0 : p.a2 + (k0 - p.k1);
    const long ldx = seg1 ? p.lda1 : p.lda2;
    const int xcols = seg1 ? p.k1 - k0 : p.k1 + p.k2 - k0;
    const float* const dbase = p.dy + n0;
    const int dcols = p.n - n0;
    // lane = channel pair: waves 0, 1 stage the 256 dY channels, waves 2, 3 the 256 x channels (wave-uniform)
    const bool isdy = wave_u < 2;
    const int pr = tid, c = isdy ? 2 * pr : 2 * pr - BBN;                       // channel inside its operand
    const int cc = isdy ? (c < dcols ? c : 0) : (c < xcols ? c : 0);            // past the edge: channel 0 (its products are never stored)
    const long ldop = isdy ? p.ldd : ldx;
#if defined(__HIP_DEVICE_COMPILE__)
    long xrec = ((long)(p.m - 1) * ldx + xcols) * 4, drec = ((long)(p.m - 1) * p.ldd + dcols) * 4;
    if (xrec > 0x7fffffffL) xrec = 0x7fffffffL;
    if (drec > 0x7fffffffL) drec = 0x7fffffffL;
    const __amdgpu_buffer_rsrc_t rop = isdy ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dbase), 0, (int)drec, 0x00020000)
                                            : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xbase), 0, (int)xrec, 0x00020000);
#endif
    // item u = token octet u of this lane's channel pair: rows 8 u .. 8 u + 7 of the step (through the scalar offset of the loads);
    // LDS channel 2 pr (2 pr + 1 sits 64 bytes on), octet slot u ^ (pr & 3)
    const int ivoff0 = cc * 4;
    int ildso[BITEMS];
#pragma unroll
    for (int u = 0; u < BITEMS; ++u) ildso[u] = 2 * pr * 64 + ((u ^ (pr & 3)) << 4);
    float bsum[BITEMS][2];
    float ld[BITEMS][2][8];                                                     // [item][channel of the pair][token]
    u32x4 hold3[BITEMS][2];                                                     // third pieces of the NEXT step, until the plane is free
#pragma unroll
    for (int u = 0; u < BITEMS; ++u) bsum[u][0] = bsum[u][1] = 0.f;
    auto load_item = [&](int u, int step) {
#if defined(__HIP_DEVICE_COMPILE__)
        const long mb = m_begin + (long)step * BMS;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rop, ivoff0, (int)((mb + 8 * u + e) * ldop * 4), 0);
            const unsigned lo = v[0], hi = v[1];
            ld[u][0][e] = __builtin_bit_cast(float, lo);
            ld[u][1][e] = __builtin_bit_cast(float, hi);
        }
#else
        (void)u; (void)step;
#endif
    };
    auto clampi = [&](int st) { return st < steps ? st : steps - 1; };
    char* const s3 = smem + 4 * BPLANE;                                         // the third-piece plane

    // fragment offsets inside a plane: x (row operand) channel 256 + wk*128 + kb*16 + l15; dY (column operand) channel wn*128 + nb*16 + l15
    // (block b of 16 channels sits 1024 bytes on; the octet swizzle only depends on l15: every block start is a multiple of 8 channels)
    const int xoff0 = (BBN + wk * 128 + l15) * 64 + ((g ^ ((l15 >> 1) & 3)) << 4);
    const int doff0 = (wn * 128 + l15) * 64 + ((g ^ ((l15 >> 1) & 3)) << 4);
    auto xoff = [&](int b) { return xoff0 + b * 1024; };
    auto doff = [&](int b) { return doff0 + b * 1024; };
    f32x4 acc[8][8];                                                            // [kb][nb]
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (steps > 0) {                                                            // prologue: step 0 whole, the rows of step 1 requested
#pragma unroll
        for (int u = 0; u < BITEMS; ++u) load_item(u, 0);
#pragma unroll
        for (int u = 0; u < BITEMS; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float (&x)[8] = ld[u][h];
                if (do_bias && isdy) bsum[u][h] += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
                bf16x8 a, b, cpl;
                split8(x, a, b, cpl);
                *reinterpret_cast<bf16x8*>(smem + ildso[u] + 64 * h) = a;
                *reinterpret_cast<bf16x8*>(smem + BPLANE + ildso[u] + 64 * h) = b;
                *reinterpret_cast<bf16x8*>(s3 + ildso[u] + 64 * h) = cpl;
            }
#pragma unroll
        for (int u = 0; u < BITEMS; ++u) load_item(u, clampi(1));
    }
    const float cnt_flag = (do_bias && isdy) ? 1.f : 0.f;
    for (int s = 0; s < steps; ++s) {
        __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): my plane writes of this step are done
        __syncthreads();                                  // everyone's are (first two planes of stage s & 1, the third-piece plane)
        char* const cur = smem + (s & 1) * 2 * BPLANE;    // planes 1, 2 of this step; plane 3 at s3
        char* const nxt = smem + ((s + 1) & 1) * 2 * BPLANE;
        const bool more = s + 1 < steps;
        const float cnt = more ? cnt_flag : 0.f;          // rows past the end are duplicates: never counted into the bias sums
        const int s2 = clampi(s + 2);
        auto plane = [&](char* base12, int pl) -> char* { return pl < 2 ? base12 + pl * BPLANE : s3; };
        // fragments of the step, in the order the first MFMAs consume them: x3 | d1 x 8 | x1 | d3 x 8 | x2 | d2 x 8
        bf16x8 df[8][3], xf[2][3];
        xf[0][2] = *reinterpret_cast<const bf16x8*>(plane(cur, 2) + xoff(0));
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) df[nb][0] = *reinterpret_cast<const bf16x8*>(plane(cur, 0) + doff(nb));
        xf[0][0] = *reinterpret_cast<const bf16x8*>(plane(cur, 0) + xoff(0));
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) df[nb][2] = *reinterpret_cast<const bf16x8*>(plane(cur, 2) + doff(nb));
        xf[0][1] = *reinterpret_cast<const bf16x8*>(plane(cur, 1) + xoff(0));
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) df[nb][1] = *reinterpret_cast<const bf16x8*>(plane(cur, 1) + doff(nb));
        __builtin_amdgcn_sched_barrier(0);
        auto kblock = [&](auto kbc) {
            constexpr int kb = decltype(kbc)::value;
            constexpr int u = kb >> 1, h = kb & 1;        // this block splits channel h of item u (the rows of step s + 1)
            constexpr int NR = kb < 7 ? 3 : 0;            // fragment reads of the next k block
            constexpr int S0 = NR, S1 = S0 + 22, S2 = S1 + 5, S3 = S2 + 2, S4 = S3 + (h ? 8 : 0), S5 = S4 + (kb == 7 ? 8 : 0);
            static_assert(S5 <= 48, "one MFMA per filler slot");
            if constexpr (kb == 7) {
                // every wave has read the last third-piece fragments (x3 of k block 7 came in under block 6): the plane is free
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __syncthreads();
            }
            unsigned su1[4], su2[4], su3[4];
            float sr0[4], sr1[4], st0[4], st1[4], bs0, bs1, bs2, bs3;
            const float (&x)[8] = ld[u][h];
            auto split_op = [&](auto nc) {
                constexpr int n = decltype(nc)::value;                      // 0..43: pairs (0,1) interleaved, then (2,3)
                constexpr int q = 2 * (n / 22) + (n & 1), o = (n % 22) >> 1;
                if constexpr (o == 0) su1[q] = cvt_pk_bf16(x[2 * q], x[2 * q + 1]);
                if constexpr (o == 1) st0[q] = bf_lo(su1[q]);
                if constexpr (o == 2) st1[q] = bf_hi(su1[q]);
                if constexpr (o == 3) sr0[q] = x[2 * q] - st0[q];
                if constexpr (o == 4) sr1[q] = x[2 * q + 1] - st1[q];
                if constexpr (o == 5) su2[q] = cvt_pk_bf16(sr0[q], sr1[q]);
                if constexpr (o == 6) st0[q] = bf_lo(su2[q]);
                if constexpr (o == 7) st1[q] = bf_hi(su2[q]);
                if constexpr (o == 8) sr0[q] = sr0[q] - st0[q];
                if constexpr (o == 9) sr1[q] = sr1[q] - st1[q];
                if constexpr (o == 10) su3[q] = cvt_pk_bf16(sr0[q], sr1[q]);
            };
            auto filler = [&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if constexpr (k < S0) {
                    constexpr int pl = k == 0 ? 2 : k == 1 ? 0 : 1;         // x3 first: the next block opens with it
                    xf[(kb + 1) & 1][pl] = *reinterpret_cast<const bf16x8*>(plane(cur, pl) + xoff(kb < 7 ? kb + 1 : 7));
                } else if constexpr (k < S1) {
                    split_op(std::integral_constant<int, 2 * (k - S0)>{});
                    split_op(std::integral_constant<int, 2 * (k - S0) + 1>{});
                } else if constexpr (k < S2) {                              // bias partial sum of this (channel, octet): same association as the other bodies
                    constexpr int j = k - S1;
                    if constexpr (j == 0) { bs0 = x[0] + x[1]; bs1 = x[2] + x[3]; }
                    if constexpr (j == 1) { bs2 = x[4] + x[5]; bs3 = x[6] + x[7]; }
                    if constexpr (j == 2) { bs0 = bs0 + bs1; bs2 = bs2 + bs3; }
                    if constexpr (j == 3) bs0 = bs0 + bs2;
                    if constexpr (j == 4) bsum[u][h] = cnt != 0.f ? bsum[u][h] + bs0 : bsum[u][h];
                } else if constexpr (k < S3) {
                    constexpr int pl = k - S2;
                    const u32x4 v = pl == 0 ? u32x4{su1[0], su1[1], su1[2], su1[3]} : u32x4{su2[0], su2[1], su2[2], su2[3]};
                    *reinterpret_cast<u32x4*>(nxt + pl * BPLANE + ildso[u] + 64 * h) = v;
                    if constexpr (pl == 1) hold3[u][h] = u32x4{su3[0], su3[1], su3[2], su3[3]};
                } else if constexpr (k < S4) {
                    // both channels of item u are split: its registers take the rows of step s + 2
#if defined(__HIP_DEVICE_COMPILE__)
                    constexpr int e = k - S3;
                    const long mb = m_begin + (long)s2 * BMS;
                    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rop, ivoff0, (int)((mb + 8 * u + e) * ldop * 4), 0);
                    const unsigned lo = v[0], hi = v[1];
                    ld[u][0][e] = __builtin_bit_cast(float, lo);
                    ld[u][1][e] = __builtin_bit_cast(float, hi);
#endif
                } else if constexpr (k < S5) {
                    constexpr int w = k - S4;                               // the eight third pieces of the next step, plane now free
                    *reinterpret_cast<u32x4*>(s3 + ildso[w >> 1] + 64 * (w & 1)) = hold3[w >> 1][w & 1];
                }
            };
            auto product = [&](auto pc) {
                constexpr int prd = decltype(pc)::value;
                constexpr int xp = prd == 0 ? 2 : (prd == 1 || prd >= 4) ? 0 : 1;                              // x3, x1, x2, x2, x1, x1
                constexpr int dp = prd == 0 ? 0 : prd == 1 ? 2 : prd == 2 ? 1 : prd == 3 ? 0 : prd == 4 ? 1 : 0;   // d1, d3, d2, d1, d2, d1
                dsc_static_for<8>([&](auto nbc) {
                    constexpr int nb = decltype(nbc)::value;
                    acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[kb & 1][xp], df[nb][dp], acc[kb][nb], 0, 0, 0);
                    filler(std::integral_constant<int, prd * 8 + nb>{});
                    __builtin_amdgcn_sched_barrier(0);
                });
            };
            dsc_static_for<6>(product);
        };
        dsc_static_for<8>(kblock);
    }
    __builtin_amdgcn_s_waitcnt(0x0070);

    float* out = p.out + (long)split * p.slab;
    if (do_bias) {
        // every dY channel has four items (token octets) in one lane: partial sums through LDS, summed in the order of the other bodies
        float* bs = reinterpret_cast<float*>(smem);
        __syncthreads();
        if (isdy) {
#pragma unroll
            for (int u = 0; u < BITEMS; ++u) { bs[u * BBN + 2 * pr] = bsum[u][0]; bs[u * BBN + 2 * pr + 1] = bsum[u][1]; }
        }
        __syncthreads();
        if (tid < BBN && n0 + tid < p.n)
            p.bias_out[(long)split * p.bias_slab + n0 + tid] = (bs[tid] + bs[BBN + tid]) + (bs[2 * BBN + tid] + bs[3 * BBN + tid]);
    }
    const bool vec = (p.ldo & 3) == 0 && (p.slab & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
        const int j = n0 + wn * 128 + nb * 16 + l15;
        if (j >= p.n) continue;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const int i = k0 + wk * 128 + kb * 16 + 4 * g;
            if (vec && i + 3 < p.kvalid) {
                *reinterpret_cast<f32x4*>(out + (long)j * p.ldo + i) = acc[kb][nb];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (i + e < p.kvalid) out[(long)j * p.ldo + i + e] = acc[kb][nb][e];
            }
        }
    }
}

}  // namespace dsc_tn_split

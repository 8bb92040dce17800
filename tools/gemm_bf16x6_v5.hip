// EXPERIMENT, second generation of tools/gemm_bf16x6.hip (same arithmetic: f32 operands split exactly into three bf16 pieces, six
// products, f32 accumulation on v_mfma_f32_16x16x32_bf16).  What the first run showed (profiles/r03_bf16x6_first_run.txt): the K loop
// already runs at the chip's power-limited bf16 rate (5.1 us per product at K = 512 = 2.0 GHz x 100 % of the MFMA issue rate), but
// 18-24 us of every launch are NOT overlapped with it -- launch, first tile, and the epilogue's HBM burst (output + residual of all
// 256 CUs at once) -- because one 8-wave block owns a CU's whole LDS (156 KiB), so nothing else can run beside it.
//
// This form makes two blocks fit on a CU so that one block's prologue / epilogue runs under the other's K loop:
//   * 4 waves per block (2 scenes x 2 channel groups: 160 tokens x 128 channels at RB = 5), 2 blocks per CU;
//   * the weight planes never touch LDS: they are stored FRAGMENT-MAJOR ([plane][16-channel block][K tile][lane][8 bf16] -- a wave's
//     fragment is one contiguous KiB) and each wave loads its 12 fragments of the next K tile straight from L2 into a second
//     register set (the weights of a layer are 1.5 MB: L2-resident);
//   * LDS holds only the token tile as three bf16 planes (split once per block by the staging threads, as PIPE 3 of the first
//     generation): 2 stages x 30 KiB per block.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 32, NW = 4, T = NW * 64, WM = 2, WN = 2, BN = 64 * WN;

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    f32x2v v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

__device__ __forceinline__ void split8(const f32x4 lo, const f32x4 hi, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
    u32x4 a, b, c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x0 = q < 2 ? lo[2 * q] : hi[2 * q - 4], x1 = q < 2 ? lo[2 * q + 1] : hi[2 * q - 3];
        const unsigned u1 = cvt_pk_bf16(x0, x1);
        const float r0 = x0 - bf_lo(u1), r1 = x1 - bf_hi(u1);
        const unsigned u2 = cvt_pk_bf16(r0, r1);
        const float s0 = r0 - bf_lo(u2), s1 = r1 - bf_hi(u2);
        a[q] = u1;
        b[q] = u2;
        c[q] = cvt_pk_bf16(s0, s1);
    }
    p1 = __builtin_bit_cast(bf16x8, a);
    p2 = __builtin_bit_cast(bf16x8, b);
    p3 = __builtin_bit_cast(bf16x8, c);
}

// w [n][k] f32 -> planes [3][n/16][k/32][64 lanes][8] bf16: lane = 16*g + l15 holds channel 16*nb + l15, k = 32*kt + 8*g .. +7
__global__ void split_planes_fm_kernel(const float* __restrict__ w, int n, int k, uint16_t* __restrict__ planes) {
    const long item = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one (channel block, k tile, lane) = 8 elements
    const int kts = k / 32;
    const long items = (long)(n / 16) * kts * 64;
    if (item >= items) return;
    const int lane = (int)(item & 63);
    const long blk = item >> 6;
    const int kt = (int)(blk % kts), nb = (int)(blk / kts);
    const int row = nb * 16 + (lane & 15), k0 = kt * 32 + 8 * (lane >> 4);
    const f32x4 lo = *(const f32x4*)(w + (long)row * k + k0), hi = *(const f32x4*)(w + (long)row * k + k0 + 4);
    bf16x8 a, b, c;
    split8(lo, hi, a, b, c);
    const long plane = (long)n * k;
    *(bf16x8*)(planes + item * 8) = a;
    *(bf16x8*)(planes + plane + item * 8) = b;
    *(bf16x8*)(planes + 2 * plane + item * 8) = c;
}

struct Args {
    const float* x; const float* x2; int k1;
    const uint16_t* planes;  // fragment-major, see above
    const float* bias; float* out;
    int m, n, k, lda, ldc;
    int ntok;
    const float* gamma; const float* beta; float eps;
    const float* scale_shift; int ld_ss; int ss_mode; const int64_t* ss_index;
    float* preact; int ld_pre;
    const float* residual; int ldr;
    int act;
    long long* stamps;       // optional [blocks][8]: wall clock (100 MHz) and shader clock at start / K loop start / K loop end / end
};

#define STAMP(i) do { if (p.stamps && threadIdx.x == 0) { p.stamps[blockIdx.x * 8 + (i)] = wall_clock64(); p.stamps[blockIdx.x * 8 + 4 + (i)] = clock64(); } } while (0)
static long long* g_stamps = nullptr;

template <int PRODUCTS, bool GN, int RB, int SCHED>
__global__ __launch_bounds__(T, 2) void gemm_split_v5_kernel(const Args p) {
    constexpr int BM = 16 * RB * WM;
    constexpr int X_PLANE = BM * BK * 2, STAGE = 3 * X_PLANE;
    constexpr int ITEMS_W = BM * 4 / NW;                 // (row, k-octet) items per wave
    constexpr int NIT = (ITEMS_W + 63) / 64;             // per lane (RB 5: 160 per wave -> 3 rounds, the last one half full)
    constexpr int DUMP = (ITEMS_W % 64) ? 1024 : 0;
    static_assert(NIT <= RB, "the splits ride in the last NIT token blocks of a tile");
    static_assert(2 * (2 * STAGE + DUMP) <= 160 * 1024, "two blocks per CU");
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE + DUMP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave_u % WM, wn = wave_u / WM;
    STAMP(0);
    const int scenes = (p.m + p.ntok - 1) / p.ntok;
    const int cbs = p.n / BN, rbs = (scenes + WM - 1) / WM;
    int rb, cb;
    if ((rbs & 7) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        rb = xcd * (rbs >> 3) + idx / cbs;
        cb = idx % cbs;
    } else {
        rb = blockIdx.x / cbs;
        cb = blockIdx.x % cbs;
    }
    const int row0 = rb * WM * p.ntok, col0 = cb * BN;
    const int rows_here = p.m - row0;
    const float* const xb = p.x + (int64_t)row0 * p.lda;
    const float* const xb2 = p.x2 ? p.x2 + (int64_t)row0 * p.lda : xb;
    const int KT = p.k / BK;
    const int g = lane >> 4, l15 = lane & 15;

    // weight fragments: plane pl, channel block (col0 + wn*64)/16 + j, K tile kt -> 1 KiB at ((pl*NB + nb)*KT + kt) * 1024
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.planes), 0, 0x7fffffff, 0x00020000);
#endif
    const int nb0 = (col0 + wn * 64) / 16, NB = p.n / 16;
    auto load_w = [&](int kt, bf16x8 (&wf)[4][3]) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const int so = __builtin_amdgcn_readfirstlane(((pl * NB + nb0 + j) * KT + kt) * 1024);   // wave-uniform: scalar offset, no waterfall loop
                wf[j][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, so, 0));
            }
#else
        (void)kt; (void)wf;
#endif
    };

    f32x4 acc[RB][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) b4 = *(const f32x4*)(p.bias + col0 + wn * 64 + j * 16 + 4 * g);
#pragma unroll
        for (int i = 0; i < RB; ++i) acc[i][j] = b4;
    }

    auto mma_block = [&](const bf16x8 (&wf)[4][3], const bf16x8& x1, const bf16x8& x2, const bf16x8& x3, f32x4 (&c)[4]) {
        if (PRODUCTS >= 6) {
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][2], x1, c[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], x3, c[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][1], x2, c[j], 0, 0, 0);
        }
        if (PRODUCTS >= 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][1], x1, c[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], x2, c[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], x1, c[j], 0, 0, 0);
    };
    constexpr int NMMA = 4 * PRODUCTS;

    // staging items of this lane: LDS row r (scene r / (16 RB), token r % (16 RB); padding rows re-read row 0), k-octet q
    int ivoff[NIT], ildso[NIT];
    bool ion[NIT];
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int idx = 64 * u + lane;
        ion[u] = idx < ITEMS_W;
        const int t = wave_u * ITEMS_W + (ion[u] ? idx : 0);
        const int r = t >> 2, q = t & 3;
        const int sc = r / (16 * RB), tk = r % (16 * RB), gr = sc * p.ntok + tk;
        ivoff[u] = (tk < p.ntok && gr < rows_here ? gr : 0) * p.lda * 4 + q * 32;
        ildso[u] = r * 64 + ((q ^ ((r >> 1) & 3)) << 4);
    }
    int xoff[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int r = (wm * RB + i) * 16 + l15;
        xoff[i] = r * 64 + ((g ^ ((r >> 1) & 3)) << 4);
    }
    f32x4 ld[NIT][2];
    auto load_items = [&](int kt) {
        const int k0 = kt * BK;
        const bool seg1 = k0 < p.k1;
        const char* base = (const char*)(seg1 ? xb : xb2) + (seg1 ? k0 : k0 - p.k1) * 4;
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            ld[u][0] = *(const f32x4*)(base + ivoff[u]);
            ld[u][1] = *(const f32x4*)(base + ivoff[u] + 16);
        }
    };
    auto store_item = [&](int u, char* stage) {
        bf16x8 a, b, c;
        split8(ld[u][0], ld[u][1], a, b, c);
        char* d0 = ion[u] ? stage + ildso[u] : smem + 2 * STAGE + lane * 16;
        const int ps = ion[u] ? X_PLANE : 0;
        *(bf16x8*)d0 = a;
        *(bf16x8*)(d0 + ps) = b;
        *(bf16x8*)(d0 + 2 * ps) = c;
    };

    // SCHED 2/3: the second block to become resident on a CU (dispatch order: the second half of the grid) runs at a higher wave
    // priority.  Two symmetric blocks walk their K loops in lock-step and reach their epilogues together (nothing overlaps);
    // with a priority the favoured block takes the matrix pipe, finishes early and streams its outputs while the other one
    // runs its K loop alone at full rate.
    if (SCHED >= 2 && blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_setprio(SCHED == 2 ? 1 : 3);
    bf16x8 wfA[4][3], wfB[4][3];
    load_items(0);
    load_w(0, wfA);
#pragma unroll
    for (int u = 0; u < NIT; ++u) store_item(u, smem);

    auto tile = [&](int kt, const bf16x8 (&wfc)[4][3], bf16x8 (&wfn)[4][3]) {
        __builtin_amdgcn_s_waitcnt(0x0070);              // vmcnt(0) lgkmcnt(0): my plane writes of tile kt and my weight fragments are here
        __syncthreads();                                 // everyone's plane writes are; nobody reads the other stage any more
        char* cur = smem + (kt & 1) * STAGE;
        char* nxt = smem + ((kt + 1) & 1) * STAGE;
        const int kn = min(kt + 1, KT - 1);              // the tail re-stages the last tile (no branch in the loop)
        // attribution probes (WRONG results, timing only): act bit 6 = token rows always from K tile 0 (L2-hot after the first
        // touch: removes the memory latency of the token stream), bit 8 = weight fragments always K tile 0
        load_items((p.act & 64) ? 0 : kn);               // issued FIRST: vmcnt(12) below == these have arrived
        __builtin_amdgcn_sched_barrier(0);               // (keeps the weight loads behind them in program order)
        load_w((p.act & 256) ? 0 : kn, wfn);
        bf16x8 xf[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) xf[0][pl] = *(const bf16x8*)(cur + pl * X_PLANE + xoff[0]);
        if (SCHED) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (i + 1 < RB) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) xf[(i + 1) & 1][pl] = *(const bf16x8*)(cur + pl * X_PLANE + xoff[i + 1]);
            }
            if (i == RB - NIT) {
                if (SCHED) __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0x0f7c);      // vmcnt(12): the staged f32 rows of the next tile have arrived (the 12 weight loads may still fly)
                if (SCHED) __builtin_amdgcn_sched_barrier(0);
            }
            if (i >= RB - NIT) store_item(i - (RB - NIT), nxt);
            mma_block(wfc, xf[i & 1][0], xf[i & 1][1], xf[i & 1][2], acc[i]);
            if (SCHED) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i + 1 < RB) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                if (i >= RB - NIT) {
#pragma unroll
                    for (int q = 0; q < NMMA - 2; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, (44 + NMMA - 3) / (NMMA - 2), 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    STAMP(1);
    int kt = 0;
    for (; kt + 1 < KT; kt += 2) {
        tile(kt, wfA, wfB);
        tile(kt + 1, wfB, wfA);
    }
    if (kt < KT) tile(kt, wfA, wfB);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    STAMP(2);

    const int cbase = col0 + wn * 64 + 4 * g;
    const int srow = wm * p.ntok;
    bool valid[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) valid[i] = i * 16 + l15 < p.ntok && srow + i * 16 + l15 < rows_here;
    float* const ob = p.out + (int64_t)(row0 + srow + l15) * p.ldc + cbase;
    const float* const rbp = p.residual ? p.residual + (int64_t)(row0 + srow + l15) * p.ldr + cbase : nullptr;
    if constexpr (!GN) {
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (valid[i]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 y = acc[i][j];
                    if (p.act == 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = 0.5f * y[e] * (1.f + erff(y[e] * 0.70710678118654752f));
                    } else if (p.act == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = y[e] / (1.f + __expf(-y[e]));
                    }
                    if (rbp) y += *(const f32x4*)(rbp + (int64_t)i * 16 * p.ldr + j * 16);
                    *(f32x4*)(ob + (int64_t)i * 16 * p.ldc + j * 16) = y;
                }
            }
        }
    } else {
        const int scene = min(row0 / p.ntok + wm, p.m / p.ntok - 1);
        const bool per_row = p.ss_mode == 1 || p.ss_mode == 3;
        f32x4 ga[4], be[4], sc[4], sh[4];
        int64_t ssrow = scene;
        if (p.ss_mode == 4) ssrow = p.ss_index[scene];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ga[j] = *(const f32x4*)(p.gamma + cbase + j * 16);
            be[j] = *(const f32x4*)(p.beta + cbase + j * 16);
            if (p.ss_mode == 2 || p.ss_mode == 4) {
                const float* ssr = p.scale_shift + ssrow * p.ld_ss + cbase + j * 16;
                sc[j] = *(const f32x4*)ssr;
                sh[j] = *(const f32x4*)(ssr + p.n);
            } else {
                sc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                sh[j] = sc[j];
            }
        }
        if (p.preact) {
            float* const pb = p.preact + (int64_t)(row0 + srow + l15) * p.ld_pre + cbase;
#pragma unroll
            for (int i = 0; i < RB; ++i)
                if (valid[i]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) *(f32x4*)(pb + (int64_t)i * 16 * p.ld_pre + j * 16) = acc[i][j];
                }
        }
        auto wave_sum = [](float v) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            return v;
        };
        const float inv_cnt = 1.f / (float)(p.ntok * 64);
        float s0 = 0.f;
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                s0 += valid[i] ? (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]) : 0.f;
        const float mean = wave_sum(s0) * inv_cnt;
        float q0 = 0.f;
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = valid[i] ? acc[i][j][e] - mean : 0.f;
                    q0 = fmaf(d, d, q0);
                }
        const float rstd = 1.f / sqrtf(wave_sum(q0) * inv_cnt + p.eps);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = rstd * ga[j][e], sp1 = sc[j][e] + 1.f;
                ga[j][e] = a * sp1;
                be[j][e] = fmaf(be[j][e] - mean * a, sp1, sh[j][e]);
            }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (valid[i]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 y, s1 = {1.f, 1.f, 1.f, 1.f}, s2 = {0.f, 0.f, 0.f, 0.f};
                    if (per_row) {
                        const int64_t row = p.ss_mode == 1 ? (int64_t)row0 + srow + i * 16 + l15 : (int64_t)i * 16 + l15;
                        const float* ssr = p.scale_shift + row * p.ld_ss + cbase + j * 16;
                        s1 = *(const f32x4*)ssr + 1.f;
                        s2 = *(const f32x4*)(ssr + p.n);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = fmaf(acc[i][j][e], ga[j][e], be[j][e]);
                        if (per_row) t = fmaf(t, s1[e], s2[e]);
                        y[e] = t / (1.f + __expf(-t));
                    }
                    if (rbp) y += *(const f32x4*)(rbp + (int64_t)i * 16 * p.ldr + j * 16);
                    *(f32x4*)(ob + (int64_t)i * 16 * p.ldc + j * 16) = y;
                }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);
    STAMP(3);
}

template <int RB>
static int launch_rb(const Args& a, int gn, int products, int sched, hipStream_t s) {
    if (a.n % BN || a.ntok > 16 * RB || a.ntok <= 0) return 2;
    if ((int64_t)WM * a.ntok * a.lda * 4 >= 0x7fffffffLL || 3LL * a.n * a.k * 2 >= 0x7fffffffLL) return 3;
    const int scenes = (a.m + a.ntok - 1) / a.ntok;
    const unsigned grid = (unsigned)(((scenes + WM - 1) / WM) * (a.n / BN));
    if (gn) {
        if (products != 6 || a.m % a.ntok) return 2;
        if (sched == 3) gemm_split_v5_kernel<6, true, RB, 3><<<grid, T, 0, s>>>(a);
        else if (sched == 2) gemm_split_v5_kernel<6, true, RB, 2><<<grid, T, 0, s>>>(a);
        else if (sched) gemm_split_v5_kernel<6, true, RB, 1><<<grid, T, 0, s>>>(a);
        else gemm_split_v5_kernel<6, true, RB, 0><<<grid, T, 0, s>>>(a);
    } else if (products == 6) {
        if (sched == 3) gemm_split_v5_kernel<6, false, RB, 3><<<grid, T, 0, s>>>(a);
        else if (sched == 2) gemm_split_v5_kernel<6, false, RB, 2><<<grid, T, 0, s>>>(a);
        else if (sched) gemm_split_v5_kernel<6, false, RB, 1><<<grid, T, 0, s>>>(a);
        else gemm_split_v5_kernel<6, false, RB, 0><<<grid, T, 0, s>>>(a);
    } else if (products == 1) gemm_split_v5_kernel<1, false, RB, 1><<<grid, T, 0, s>>>(a);
    else if (products == 3) gemm_split_v5_kernel<3, false, RB, 1><<<grid, T, 0, s>>>(a);
    else return 2;
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace

extern "C" {

int bf16x6_layout_fragment_major(void) { return 1; }
void bf16x6_set_stamps(long long* dev_ptr) { g_stamps = dev_ptr; }

// count = n * k; n % 16 == 0, k % 32 == 0
int bf16x6_split_planes_nk(const float* w, int n, int k, uint16_t* planes, hipStream_t s) {
    if (n % 16 || k % 32) return 2;
    const long items = (long)(n / 16) * (k / 32) * 64;
    split_planes_fm_kernel<<<(unsigned)((items + 255) / 256), 256, 0, s>>>(w, n, k, planes);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// same signature as tools/gemm_bf16x6.hip's launcher; `pipe`: 0 = compiler-scheduled, anything else = pinned schedule; `tile` unused
int bf16x6_launch(const float* x, const float* x2, int k1, int lda, const uint16_t* planes, const float* bias, float* out, int ldc,
                  int m, int n, int k, int act, const float* residual, int ldr, int gn, const float* gamma, const float* beta,
                  float eps, const float* scale_shift, int ld_ss, int ss_mode, const int64_t* ss_index, float* preact, int ld_pre, int ntok,
                  int products, int pipe, int tile, hipStream_t s) {
    (void)tile;
    if (k % BK || k1 % BK || k1 <= 0 || k1 > k || (k1 < k && !x2) || (lda & 3) || (ldc & 3) || (ldr & 3) || (ld_ss & 3) || m <= 0)
        return 2;
    if (gn && (ntok <= 0 || !gamma || !beta || ss_mode < 0 || ss_mode > 4 || (ss_mode && !scale_shift) || (ss_mode == 4 && !ss_index) ||
               (ld_pre & 3)))
        return 2;
    const bool small = ntok > 0 && ntok <= 32;
    Args a{x, k1 < k ? x2 : nullptr, k1, planes, bias, out, m, n, k, lda, ldc, small ? ntok : (ntok > 0 ? ntok : 80),
           gamma, beta, eps, scale_shift, ld_ss, scale_shift ? ss_mode : 0, ss_index, preact, ld_pre, residual, ldr, act, g_stamps};
    if (small) return launch_rb<2>(a, gn, products, pipe, s);
    return launch_rb<5>(a, gn, products, pipe, s);
}

}  // extern "C"

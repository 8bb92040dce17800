// Shared device helpers for libdiffuscene_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/diffuscene_hip.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DSC_WAVE 64

#include <utility>
template <class F, int... I>
__device__ __forceinline__ void dsc_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), in order: compile-time indices for hand-placed instruction streams
template <int N, class F>
__device__ __forceinline__ void dsc_static_for(F&& f) { dsc_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// hipGetLastError() is per-thread and sticky: another library (PyTorch probing devices while it initialises) can leave
// an unrelated error behind.  Flush it before our launch so DSC_LAUNCH_CHECK reports only this launch.
#define DSC_CLEAR_STALE_ERROR() ((void)hipGetLastError())

#define DSC_LAUNCH_CHECK()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return (int)e__;              \
    } while (0)

__device__ __forceinline__ float dsc_silu(float x) { return x / (1.0f + expf(-x)); }
// epilogue variant: hardware exp2/rcp (v_exp_f32, v_rcp_f32; ~1 ulp each), keeps the GEMM epilogue off the VALU critical path.
// (__frcp_rn would expand to the correctly rounded ~10-instruction division sequence.)
__device__ __forceinline__ float dsc_silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// erf(x), branch-free: Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute, i.e. fp32 rounding level next to the 1 it is
// added to in GELU) -- one v_rcp + one v_exp + 8 FMAs instead of the two-branch libm erff (both branches execute in a
// divergent wave: ~35 VALU per element; measured 29 us of a 187 us n = 1024 GEMM launch with the GELU epilogue).
__device__ __forceinline__ float dsc_erf(float x) {
    const float a = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.0f - p * t * __expf(-a * a);
    return copysignf(e, x);
}
// nn.GELU() default: 0.5 x (1 + erf(x / sqrt(2)))
__device__ __forceinline__ float dsc_gelu(float x) { return 0.5f * x * (1.0f + dsc_erf(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float dsc_act(float x, int act) {
    if (act == DSC_ACT_GELU) return dsc_gelu(x);
    if (act == DSC_ACT_SILU) return dsc_silu(x);
    if (act == DSC_ACT_LEAKY01) return x > 0.0f ? x : 0.1f * x;
    return x;
}

// d act(x) / dx (dsc_activation_bwd_f32 and the actgrad_x epilogue of the split GEMM share it: same values either way).  In a GEMM
// epilogue this arithmetic is EXPOSED (every wave does it at the same moment, after the K loop), so it is kept short: GELU' = Phi(x) +
// x phi(x) takes its Gaussian density from the exponential the erf approximation computes anyway (exp(-(x/sqrt2)^2) = exp(-x^2/2)) --
// one v_rcp, one v_exp, ~14 FMAs; SiLU' uses the hardware reciprocal.  (Round-4 first form: libm expf twice + a division, ~45 VALU
// per element: the fused launches lost to the epilogue what the removed activation-backward launches had saved.)
__device__ __forceinline__ float dsc_act_grad(float xv, int act) {
    if (act == DSC_ACT_GELU) {
        const float a = fabsf(xv) * 0.70710678118654752440f;
        const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
        float p = fmaf(1.061405429f, t, -1.453152027f);
        p = fmaf(p, t, 1.421413741f);
        p = fmaf(p, t, -0.284496736f);
        p = fmaf(p, t, 0.254829592f);
        const float e = __expf(-a * a);
        const float cdf = 0.5f * (1.0f + copysignf(1.0f - p * t * e, xv));
        return fmaf(xv * 0.39894228040143267794f, e, cdf);
    }
    if (act == DSC_ACT_SILU) {
        const float sig = __builtin_amdgcn_rcpf(1.0f + __expf(-xv));
        return sig * fmaf(xv, 1.0f - sig, 1.0f);
    }
    if (act == DSC_ACT_LEAKY01) return xv > 0.0f ? 1.0f : 0.1f;
    return 1.0f;
}

// Block size of the softmax-attention kernels (forward: blocks.hip, backward: train.hip): four lanes per token, one pass over the
// tokens whenever they fit 512 threads (N = 80: 320 threads, five waves; N <= 64: 256 as before), host side only.
static inline unsigned dsc_attention_threads(int n) {
    if (n <= 64) return 256;
    const int waves = (n + 15) / 16;
    return waves >= 8 ? 512u : 64u * (unsigned)waves;
}

// butterfly reductions over the 64 lanes of a wave
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// wave-wide sum on the DPP data path (VALU row shifts / broadcasts, no LDS crossbar round trips): the total lands in lane 63
// and is broadcast with a readlane.  Same association order for every lane -> deterministic.
__device__ __forceinline__ float wave_sum_dpp(float v) {
    auto dpp_add = [](float x, auto ctrl, auto row_mask, auto bank_mask) {
        const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, decltype(row_mask)::value,
                                                  decltype(bank_mask)::value, true);
        return x + __builtin_bit_cast(float, y);
    };
    using std::integral_constant;
    v = dpp_add(v, integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{});   // row_shr:1
    v = dpp_add(v, integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{});   // row_shr:2
    v = dpp_add(v, integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{});   // row_shr:4
    v = dpp_add(v, integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{});   // row_shr:8
    v = dpp_add(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{}, integral_constant<int, 0xf>{});   // row_bcast:15
    v = dpp_add(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{}, integral_constant<int, 0xf>{});   // row_bcast:31
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// gemm_split.hip: the split-bf16 fast path of dsc_gemm_f32 / dsc_gemm_gn_silu_f32 (takes the launch when args->w_planes is set and
// the shape qualifies)
#define DSC_SPLIT_NOT_TAKEN (-1000)
int dsc_gemm_try_split(const dsc_gemm_args* a, bool gn, hipStream_t s);

// train.hip: deterministic slab reduction of a grouped weight-gradient launch (used by both arithmetics)
int dsc_launch_reduce_grouped(const dsc_tn_group* groups_dev, int count, int total_tiles, int splits, const float* workspace, hipStream_t s);

// Out-of-range device indices (a timestep outside the schedule tables, a gather row outside its table): every kernel that indexes a
// table with a DEVICE value clamps it into range -- memory-safe whatever the caller left in the vector (after a finished hipGraph
// reverse loop the in-graph timestep is -1) -- and the elementwise DDPM kernels also count the event in a per-translation-unit
// counter that dsc_device_error_count() sums (tests read it; a correct run leaves it at 0).  Units that use it define
// DSC_BAD_INDEX_COUNTER before including this header.
#ifdef DSC_BAD_INDEX_COUNTER
static __device__ unsigned int dsc_bad_index_count = 0;
// `first`: true for ONE thread per checked value (the caller's first thread of the first block that reads it), so the counter reads
// as "out-of-range values seen", not threads x blocks per event.
__device__ __forceinline__ int64_t dsc_checked_index(int64_t v, int64_t n, bool first = true) {
    if (v < 0 || v >= n) {
        if (first) atomicAdd(&dsc_bad_index_count, 1u);
        return (v < 0 || n < 1) ? 0 : n - 1;
    }
    return v;
}
static inline unsigned dsc_read_bad_index_count(bool reset) {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(dsc_bad_index_count), sizeof(v)) != hipSuccess) return 0xffffffffu;
    if (reset && v) {
        const unsigned z = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(dsc_bad_index_count), &z, sizeof(z));
    }
    return v;
}
#endif
// per-unit readers behind dsc_device_error_count (diffusion.hip)
unsigned dsc_bad_index_diffusion(bool reset);
unsigned dsc_bad_index_train(bool reset);
// clamp only (GEMM prologues: no counter, no branch)
__device__ __forceinline__ int64_t dsc_clamp_index(int64_t v, int64_t n) { return (v < 0 || n < 1) ? 0 : (v >= n ? n - 1 : v); }    // (n < 1: row 0, never row -1)

static inline bool dsc_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

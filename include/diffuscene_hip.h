/*
 * diffuscene_hip.h -- C ABI of libdiffuscene_hip.so (gfx950 / MI355X).
 *
 * The reference (tangjiapeng/DiffuScene) has no FFI on this path: its DDPM hot path is pure
 * PyTorch (scene_synthesis/networks/denoise_net.py, diffusion_ddpm.py, loss.py).  The drop-in
 * boundary is therefore the Python class surface; this header is the
 * thin C layer underneath it.  Each entry point names the reference code it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 unless stated; the caller (PyTorch) owns all
 *    memory, the library allocates nothing and keeps no global state;
 *  - activations are token-major: row = b * N + n (scene b, object slot n), channels contiguous.
 *    The reference's (B, C, N) conv layout is never materialised;
 *  - every call enqueues on `stream` and returns immediately (capturable in a hipGraph);
 *  - return value: 0 on success, a negative DSC_E* code for rejected arguments, or a positive
 *    hipError_t from the launch.  Nothing is silently clamped.
 */
#ifndef DIFFUSCENE_HIP_H
#define DIFFUSCENE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dsc_stream_t; /* hipStream_t */

#define DSC_EINVAL   (-1)  /* bad shape / null pointer */
#define DSC_EALIGN   (-2)  /* pointer or leading dimension not 16-byte aligned where required */
#define DSC_ERANGE   (-3)  /* size outside what the kernel supports (e.g. tokens_per_scene > 160) */

#define DSC_ACT_NONE 0
#define DSC_ACT_GELU 1   /* exact erf GELU (nn.GELU default) */
#define DSC_ACT_SILU 2
#define DSC_ACT_LEAKY01 3 /* nn.LeakyReLU(0.1): the condition MLPs of diffusion_scene_layout_ddpm.py:94-125 (elementwise kernels) */

#define DSC_SS_NONE      0
#define DSC_SS_PER_TOKEN 1   /* scale_shift row = token           (context-conditioned ResnetBlock) */
#define DSC_SS_PER_SCENE 2   /* scale_shift row = token / N       (time-conditioned ResnetBlock)    */
#define DSC_SS_PER_SLOT  3   /* scale_shift row = token % N       (instance embedding shared over B)*/
#define DSC_SS_BY_INDEX  4   /* scale_shift row = ss_index[token / N]  (per-timestep table while sampling) */

#define DSC_HEADS    4
#define DSC_DIM_HEAD 32

int dsc_version(void);

/* ---------------------------------------------------------------------------------------------
 * fp32 MFMA GEMM with fused epilogues:  Y = epilogue( [A1 | A2] . W^T + bias )
 *
 * Replaces every 1x1 nn.Conv1d / nn.Linear on the path (denoise_net.py:163,183,188,214,217,244,
 * 245,397,419-421,440,459,487-503) including the torch.cat of skip connections (:562,:566,:573),
 * which becomes the second K segment.  Arithmetic: v_mfma_f32_32x32x2_f32, exact fp32.
 *   k1, k2 multiples of 32; a1/a2/w rows 16-byte aligned.  y/residual may have any alignment.
 *   batch > 1 launches a grouped GEMM: group z uses pointers advanced by the s* element strides.
 * ------------------------------------------------------------------------------------------- */
typedef struct dsc_gemm_args {
    const float* a1; int64_t lda1; int32_t k1;
    const float* a2; int64_t lda2; int32_t k2;   /* k2 = 0 -> single segment */
    const float* w;  int64_t ldw;                /* [n][k1+k2] */
    const float* bias;                           /* [n] or NULL */
    const float* residual; int64_t ldr;          /* added after act_out, or NULL */
    float* y; int64_t ldy;
    int32_t m, n;
    int32_t act_in;                              /* reserved, must be DSC_ACT_NONE (apply dsc_activation_f32 to A first) */
    int32_t act_out;
    int32_t batch; int64_t sa1, sa2, sw, sbias, sres, sy;
    /* ---- GroupNorm epilogue (dsc_gemm_gn_silu_f32 only) ---- */
    const float* gamma; const float* beta; float eps;
    int32_t tokens_per_scene;                    /* N: GroupNorm reduces over 64 channels x N tokens (:164) */
    const float* scale_shift; int64_t ld_ss; int32_t ss_mode; /* row layout [scale(n) | shift(n)] */
    float* preact; int64_t ld_preact;            /* optional: pre-norm conv output z = [A1|A2].W^T + bias, saved for backward */
    const int64_t* ss_index;                     /* DSC_SS_BY_INDEX: device int64 per scene (the timestep vector t) */
    /* ---- optional: W pre-split into three bf16 planes [3][n][k1+k2] by dsc_split_bf16x3_f32 (same values as w) ----
     * (batch > 1: the planes [3][batch n][k] of the stacked weights, sw == n k.)
     * When set (and n % 128 == 0, 16-byte aligned y / bias / residual), dsc_gemm_f32 and dsc_gemm_gn_silu_f32 compute
     * the SAME f32 product on the bf16 matrix cores: both operands split exactly into 3 bf16 pieces, the 6 significant piece
     * products accumulated in f32 (error vs f64 <= the exact-f32 MFMA path's, ~1.5x faster).  NULL, an unsupported shape, or
     * dsc_get_gemm_arithmetic() == 0: the exact-f32 MFMA kernel runs.
     * Operand range of the split arithmetic (tests/test_gpu_split.py holds these): finite operands with |x| <= 3.3895e38 (the largest
     * bf16) give the f32 product at any magnitude and any mix of magnitudes inside a K row; pieces below the bf16 subnormal range
     * (third pieces of |x| < ~1e-33) may be flushed by the matrix cores: an absolute error <= 2^-16 |x w| on such terms, invisible
     * next to any normal-range term.  An operand that is +-inf, NaN or above the largest bf16 makes its first piece inf and its
     * residual pieces NaN, and a product that overflows f32 may meet an oppositely signed piece product: the affected outputs are
     * non-finite in both arithmetics, but NaN where the exact-f32 kernel may give +-inf. */
    const uint16_t* w_planes;
    /* ---- training-step epilogues of dsc_gemm_f32, implemented by the split-bf16 kernel only (a launch that sets them and does not
     * qualify for that kernel -- dsc_gemm_arithmetic() == 0 -- fails with DSC_EINVAL, it is never silently ignored):
     *   preact (above) with act_out != NONE : the pre-activation u = [A1|A2].W^T + bias is ALSO stored there, y = act_out(u)
     *                                         (Linear + GELU / SiLU of the encoder / decoder MLPs in one launch, u kept for backward)
     *   actgrad_x != NULL                   : y = ([A1|A2].W^T) * act_out'(actgrad_x[row][col]) (+ residual): the input-gradient GEMM
     *                                         of the NEXT layer applies the derivative of the activation whose saved pre-activation
     *                                         is actgrad_x (act_out names the activation; it is not applied to y) */
    const float* actgrad_x; int64_t ld_actgrad;
    /* DSC_SS_BY_INDEX: rows of the scale_shift table (>= 1).  Every gathered index is clamped into [0, ss_rows) on the device, so a
     * timestep vector left out of range (e.g. -1 after the last step of a captured reverse loop) cannot read outside the table. */
    int32_t ss_rows;
    /* Layout of w_planes (round 6): DSC_PLANES_ROWMAJOR = [3][n][K] (the block-staged split kernel stages them through LDS);
     * DSC_PLANES_FRAGMENT = [3][n/16][K/32][64 lanes][8]: one 1 KiB piece per (16-channel block, K tile, plane) in MFMA lane order, read
     * straight into registers by the wave-autonomous kernel (csrc/gemm_split_wave.h).  Which one a launch wants is the library's
     * decision -- ask dsc_gemm_planes_layout() BEFORE making the planes; a launch whose planes have the other layout fails with
     * DSC_EINVAL (it is never computed on the wrong bytes, never silently sent to another kernel). */
    int32_t w_planes_layout;
    /* ---- GroupNorm-backward epilogue of dsc_gemm_f32 (round 6; wave-autonomous kernel only: DSC_TILE_WAVE_DENSE with rows = whole scenes of 65..80 tokens) ----
     * The product P = [A1|A2].W^T is the gradient w.r.t. the OUTPUT of a fused Block (dsc_gemm_gn_silu_f32) whose saved pre-norm activation is gnb_z; y receives
     * the gradient w.r.t. that pre-norm activation -- what dsc_gn_silu_bwd_f32 computes from (gnb_z, P) -- and the per-scene partial sums go to gnb_dgamma /
     * gnb_dbeta / gnb_dbias (row b at + b * gnb_pstride) and, with DSC_SS_PER_SCENE, the per-scene d(scale, shift) to gnb_dss ([scenes][2 n] rows of ld_gnb_dss).
     * gamma, beta, eps, tokens_per_scene, scale_shift, ld_ss, ss_mode describe that Block (ss_mode DSC_SS_NONE or DSC_SS_PER_SCENE); bias, act_out, residual,
     * preact and actgrad_x must be unset.  A launch that sets gnb_z and does not qualify fails with DSC_EINVAL (dsc_gemm_split_tile tells beforehand). */
    const float* gnb_z; int64_t ld_gnb_z;
    float* gnb_dgamma; float* gnb_dbeta; float* gnb_dbias; int64_t gnb_pstride;
    float* gnb_dss; int64_t ld_gnb_dss;
    /* DSC_GEMM_ROW_INVARIANT: every output row must come out exactly as in a launch of the same product with any other number of rows --
     * a table built once with m = T rows stands in for per-step launches with m = B rows (the tabulated time MLP of the reverse loops,
     * engine.ss_table), and a captured loop must reproduce the eager loop bit for bit.  The launch then stays on the exact-f32 tile
     * kernels, whose K order does not depend on m (the K-parallel kernel for small launches associates the K sum differently). */
    int32_t flags;
} dsc_gemm_args;
#define DSC_GEMM_ROW_INVARIANT 1
#define DSC_PLANES_ROWMAJOR 0
#define DSC_PLANES_FRAGMENT 1

int dsc_gemm_f32(const dsc_gemm_args* args, dsc_stream_t stream);

/* Exact 3-way bf16 split of f32 matrices (x = x1 + x2 + x3, round-to-nearest at each step) for dsc_gemm_args.w_planes:
 *   planes[p][r][c], p = 0..2, each plane [rows][cols] bf16 (or [cols][rows] with transpose != 0: the planes of w^T, the weight
 * operand of the input-gradient GEMM dA = dY . W).  Output columns % 8 == 0.  items: HOST array, at most DSC_WS_MAX per call. */
/* Which arithmetic dsc_gemm_f32 (gn = 0) / dsc_gemm_gn_silu_f32 (gn != 0) would use for this launch: 1 = the split-bf16 kernel (planes
 * supplied, shape / alignment covered, launch large enough to fill the chip, DSC_GEMM != f32), 0 = the exact-f32 MFMA kernel. */
int dsc_gemm_arithmetic(const dsc_gemm_args* args, int32_t gn);

/* Which TILE of the split-bf16 kernel the launch would run on (-1: none, the exact-f32 MFMA kernel takes it).  Tests use it to say
 * which tile class a golden comparison exercised: the headline batch (B = 256, N = 80) runs the eight-wave GroupNorm tile, half of
 * it the four-wave one -- same K order and product order per output element, bit-identical results (tests/test_gpu_split.py). */
#define DSC_TILE_GN_32       0   /* 4 scenes of <= 32 tokens x 128 channels, 8 waves */
#define DSC_TILE_GN_80_W8    1   /* 2 scenes of 65..80 tokens x 256 channels, 8 waves */
#define DSC_TILE_GN_80_W4    2   /* 2 scenes of 65..80 tokens x 128 channels, 4 waves */
#define DSC_TILE_GN_48       3
#define DSC_TILE_GN_64       4
#define DSC_TILE_160x256     5   /* dense rows, 8 waves */
#define DSC_TILE_256x128     6
#define DSC_TILE_128x128     7
#define DSC_TILE_160x128_W4  8   /* dense rows, 4 waves */
#define DSC_TILE_64x256      9
#define DSC_TILE_WAVE_GN     10  /* wave-autonomous kernel: one scene of 17..80 tokens x 128 channels per wave, 4 waves per block */
#define DSC_TILE_WAVE_DENSE  11  /* the same on dense rows (groups of 80) */
#define DSC_TILE_WAVE_GN_64  12  /* half-size GroupNorm launches: one scene of 65..80 tokens x 64 channels per wave */
int dsc_gemm_split_tile(const dsc_gemm_args* args, int32_t gn);

/* The planes layout the launch wants: DSC_PLANES_ROWMAJOR / DSC_PLANES_FRAGMENT, or -1 when it stays on the exact-f32 kernel whatever
 * planes it is given (w_planes / w_planes_layout of `args` are not read). */
int dsc_gemm_planes_layout(const dsc_gemm_args* args, int32_t gn);

/* The wave-autonomous kernel family of the split arithmetic: 1 = used wherever a launch qualifies (default), 0 = never (every split
 * launch on the block-staged kernel: the round-3..5 behaviour).  Initial value from DSC_WAVE ("", "1", "auto" -> 1; "0" -> 0; anything
 * else -> DSC_EINVAL from get and from every GEMM launch).  Both families compute bit-identical results (tests/test_gpu_split.py). */
int dsc_get_split_wave(void);
int dsc_set_split_wave(int32_t mode);

/* The K-parallel exact-f32 kernel for launches too small to fill the chip (csrc/gemm_skinny.h, round 6): a block owns <= 32 token rows
 * (whole scenes) x 64 channels and splits K over its eight waves.  It takes a launch of dsc_gemm_f32 / dsc_gemm_gn_silu_f32 that stays on
 * the exact-f32 arithmetic when (k1 + k2) % 64 == 0, n % 64 == 0, scenes of <= 32 tokens, 16-byte aligned rows and ALL blocks fit one
 * round of the chip (the one-scene generation call of scripts/generate_diffusion.py:314-323, batches of a few scenes) -- same
 * products, the K sum associated as eight slice sums.  dsc_gemm_skinny: 0 = a tile kernel (or the split family) takes the launch,
 * 1 = this kernel with blocks of <= 32 rows, 2 = with blocks of <= 16 rows (v_mfma_f32_16x16x4_f32 tiles: scenes of <= 16 tokens, K % 512 == 0).  Switch: 1 = on (default), 0 = off; initial value from DSC_SKINNY ("0" -> off). */
int dsc_gemm_skinny(const dsc_gemm_args* args, int32_t gn);
int dsc_get_skinny(void);
int dsc_set_skinny(int32_t on);   /* returns the previous setting */

/* The arithmetic switch of dsc_gemm_f32 / dsc_gemm_gn_silu_f32 / the grouped weight-gradient launch chosen by the host code -- the ONE
 * source of truth (the Python engine, the training plan and bench.py ask this function, nothing else parses the environment):
 *   1 = split-bf16 wherever a launch qualifies (default), 0 = exact-f32 MFMA everywhere.
 * Initial value from the environment, strictly: DSC_GEMM unset, "" or "split" -> 1; "f32" -> 0; any other value -> get returns
 * DSC_EINVAL and every GEMM launch fails with DSC_EINVAL.  set(mode) switches it per call, process-wide (0 / 1, else DSC_EINVAL);
 * launches already captured in a hipGraph keep the kernels they were captured with. */
int dsc_get_gemm_arithmetic(void);
int dsc_set_gemm_arithmetic(int32_t mode);

/* transpose: bit 0 = planes of w^T; bit 1 (DSC_SPLIT_FRAGMENT) = fragment-major output (DSC_PLANES_FRAGMENT; output rows % 16 == 0,
 * output columns % 32 == 0) */
#define DSC_SPLIT_TRANSPOSE 1
#define DSC_SPLIT_FRAGMENT  2
typedef struct dsc_split_item { const float* w; int64_t ldw; int32_t rows, cols; uint16_t* planes; int32_t transpose; } dsc_split_item;
int dsc_split_bf16x3_f32(const dsc_split_item* items, int32_t count, dsc_stream_t stream);

/* Block.forward (denoise_net.py:167-176) as ONE kernel:
 *   Y = SiLU( GroupNorm8( [A1|A2].W^T + bias ) * (scale + 1) + shift ) (+ residual)
 * W must already be weight-standardised (dsc_weight_standardize_f32).  n must be a multiple of 128,
 * groups are 64 channels wide (nn.GroupNorm(8, 512)), m a multiple of tokens_per_scene <= 160. */
int dsc_gemm_gn_silu_f32(const dsc_gemm_args* args, dsc_stream_t stream);

/* Split-K form of dsc_gemm_f32 for short, deep products (few rows, K >= 1024): the K range is cut into `splits` equal parts
 * (k1 % (32 * splits) == 0) computed as the batch dimension of one launch into workspace slabs [splits][m][n], then summed in a
 * fixed order with bias / residual applied.  Restrictions: batch == 1, k2 == 0, act_out == DSC_ACT_NONE. */
int dsc_gemm_splitk_f32(const dsc_gemm_args* a, int32_t splits, float* workspace, int64_t workspace_floats, dsc_stream_t stream);

/* y = LayerNorm over the n = 512 output channels of ([a1 | a2] @ w^T + bias), times gamma, plus residual (optional): the
 * out-projection + LayerNorm (+ residual) of LinearAttention (denoise_net.py:216-235 with :93-102; biased variance, eps from the
 * struct, gain only).  Same argument struct as dsc_gemm_f32 (gamma = the LayerNorm gain, beta unused); one K row stride. */
int dsc_gemm_layernorm_f32(const dsc_gemm_args* a, dsc_stream_t stream);

/* Small-K linear for un-aligned inputs (first layer of _encoder_mlp on slices of the (B,N,C)
 * tensor, denoise_net.py:487,513-524; init_conv of the 5-channel re-arrangement model, :397):
 *   y[m][n] = act( sum_k x[m*ldx + k] * w[n*ldw + k] + bias[n] ),  k_in <= 64. */
int dsc_linear_smallk_f32(const float* x, int64_t ldx, int32_t k_in, const float* w, int64_t ldw,
                          const float* bias, float* y, int64_t ldy, int32_t m, int32_t n,
                          int32_t act_out, dsc_stream_t stream);
/* The same for up to DSC_SMALLK_MAX heads in ONE launch (the first layers of the per-attribute encoders, denoise_net.py:513-524:
 * every head reads its own column slice of the scene tensor and writes its own [m][n] block); items: HOST array.  Bit-identical to
 * `count` single launches. */
#define DSC_SMALLK_MAX 4
typedef struct dsc_smallk_item { const float* x; int64_t ldx; int32_t k_in; const float* w; int64_t ldw; const float* bias;
                                 float* y; int64_t ldy; } dsc_smallk_item;
int dsc_linear_smallk_grouped_f32(const dsc_smallk_item* items, int32_t count, int32_t m, int32_t n, int32_t act_out,
                                  dsc_stream_t stream);
/* dst[r][dst_col + c] = src[r][src_col + c] for c < width, over up to DSC_SMALLK_MAX column spans (HOST array): the padded outputs of
 * the stacked decoder output heads -> their columns of the (M, C) scene tensor. */
typedef struct dsc_col_span { int32_t src_col, dst_col, width; } dsc_col_span;
int dsc_gather_columns_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int32_t rows, const dsc_col_span* spans,
                           int32_t count, dsc_stream_t stream);

/* WeightStandardizedConv2d.forward weight path (denoise_net.py:84-89):
 *   out[o][:] = (w[o][:] - mean_o) * rsqrt(var_o + eps), biased variance over the row.
 * Batched: up to DSC_WS_MAX matrices per launch. */
#define DSC_WS_MAX 64
typedef struct dsc_ws_item { const float* w; float* out; int32_t rows; int32_t cols; } dsc_ws_item;
int dsc_weight_standardize_f32(const dsc_ws_item* items, int32_t count, float eps, dsc_stream_t stream);

/* LayerNorm over channels with gain (denoise_net.py:98-102), optionally + residual (Residual, :39-45):
 *   y[r][:] = (x[r][:] - mean_r) * rsqrt(var_r + eps) * g[:] (+ residual[r][:]);  d must be 512. */
int dsc_layernorm_f32(const float* x, int64_t ldx, const float* g, const float* residual, int64_t ldr,
                      float* y, int64_t ldy, int32_t m, int32_t d, float eps, dsc_stream_t stream);

/* LinearAttention / LinearAttentionCross core (denoise_net.py:226-234, :288-296), 4 heads x 32:
 *   q <- softmax over the 32 head channels, * scale; k <- softmax over the nk tokens of the scene;
 *   ctx[d][e] = sum_j k[j][d] v[j][e];  out[i][e] = sum_d ctx[d][e] q[i][d].
 * q rows b*nq+i (ldq), k/v rows b*nk+j (ldk/ldv); head h uses columns [32h, 32h+32). nq, nk <= 160. */
int dsc_linear_attention_f32(const float* q, int64_t ldq, const float* k, int64_t ldk,
                             const float* v, int64_t ldv, float* out, int64_t ldo,
                             int32_t scenes, int32_t nq, int32_t nk, float scale, dsc_stream_t stream);

/* Attention core (denoise_net.py:252-258): softmax_j( scale * q_i . k_j ) v_j per scene and head; n <= 160. */
int dsc_attention_f32(const float* q, int64_t ldq, const float* k, int64_t ldk,
                      const float* v, int64_t ldv, float* out, int64_t ldo,
                      int32_t scenes, int32_t n, float scale, dsc_stream_t stream);

/* SinusoidalPosEmb (denoise_net.py:132-139): out[b] = table[t[b]] when 0 <= t[b] < table_rows (the
 * table is computed by the host with the reference's own fp32 expression), else computed in place
 * from freq[dim/2] as [sin(t f) | cos(t f)]. */
int dsc_time_embedding_f32(const int64_t* t, int32_t b, int32_t dim, const float* table, int32_t table_rows,
                           const float* freq, float* out, dsc_stream_t stream);

/* y = act(x) elementwise (SiLU on the conditioning before ResnetBlock.mlp's Linear, :181-184). */
int dsc_activation_f32(const float* x, float* y, int64_t count, int32_t act, dsc_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Diffusion process (diffusion_ddpm.py).  Coefficient tables live on the device (the reference
 * re-uploads them on every call, :220,:232,:284); t is int64 on the device; `inner` = N*C.
 * These kernels are compiled with -ffp-contract=off and reproduce the reference's fp32
 * expression order bit for bit.
 * num_timesteps = rows of the coefficient tables: every kernel clamps the DEVICE value t[b] into [0, num_timesteps) before it
 * indexes a table (reference: the tables are indexed by torch.gather, which raises on such a t; here the launch is asynchronous,
 * so the access is made safe and the event is counted -- dsc_device_error_count).
 * ------------------------------------------------------------------------------------------- */

/* Out-of-range device timesteps the kernels below (and dsc_ddpm_loss_f32) had to clamp since the last reset; 0 in a correct run.
 * Synchronises the device (a test / debugging facility); -1 on a HIP error.  reset != 0 clears the counters. */
int64_t dsc_device_error_count(int32_t reset);

/* q_sample (:276-286) and, if v_out != NULL, _predict_v (:230-234) in one pass:
 *   x_t = sqrt_ac[t] * x0 + sqrt_1mac[t] * noise ;  v = sqrt_ac[t] * noise - sqrt_1mac[t] * x0 */
int dsc_q_sample_f32(const float* x0, const float* noise, const int64_t* t,
                     const float* sqrt_ac, const float* sqrt_1mac,
                     float* x_t, float* v_out, int32_t b, int64_t inner, int32_t num_timesteps, dsc_stream_t stream);

/* p_mean_variance + p_sample for model_var_type 'fixedsmall' (:242-264, :305-352):
 *   x0 = (mean_type v)   ca[t]*x_t - cb[t]*model_out      ca=sqrt_ac,        cb=sqrt_1mac
 *        (mean_type eps) ca[t]*x_t - cb[t]*model_out      ca=sqrt_recip_ac,  cb=sqrt_recipm1_ac
 *        (mean_type x0)  model_out
 *   clamp to [-1,1] if clip; mean = coef1[t]*x0 + coef2[t]*x_t; out = mean + (t != 0) * sigma[t] * noise
 * sigma[t] = exp(0.5 * posterior_log_variance_clipped[t]) is tabulated by the host. */
#define DSC_MEAN_EPS 0
#define DSC_MEAN_X0  1
#define DSC_MEAN_V   2
int dsc_p_sample_f32(const float* x_t, const float* model_out, const float* noise, const int64_t* t,
                     const float* ca, const float* cb, const float* coef1, const float* coef2,
                     const float* sigma, float* out, float* x0_out /* may be NULL */,
                     int32_t mean_type, int32_t clip, int32_t b, int64_t inner, int32_t num_timesteps,
                     dsc_stream_t stream);

/* In-graph timestep bookkeeping for the captured reverse loop (:365-366): t[i] += delta. */
int dsc_add_scalar_i64(int64_t* t, int32_t count, int64_t delta, dsc_stream_t stream);

/* Post-filter of generated scenes (delete_empty_from_network_samples, diffusion_scene_layout_ddpm.py:351-406): slot i is
 * dropped when samples[.., i, empty_col] >= 0 (and keep_empty == 0).  mode 0: the decision of batch row 0 is applied to every
 * scene (the reference's loop, :379); mode 1: per scene.  packed (b, n, c): kept rows first, original order, zero tail;
 * counts[b] = rows kept.  n <= 192; samples and packed must not alias. */
int dsc_postfilter_compact_f32(const float* samples, int32_t b, int32_t n, int32_t c, int32_t empty_col, int32_t mode,
                               int32_t keep_empty, float* packed, int32_t* counts, dsc_stream_t stream);

/* Inpainting overwrite of p_sample_loop_complete (:462-466): rows [0,p) of every scene of x
 * (b, n, c) are replaced by q_sample(partial, t, noise) (partial/noise are (b, p, c)). */
int dsc_complete_overwrite_f32(float* x, const float* partial, const float* noise, const int64_t* t,
                               const float* sqrt_ac, const float* sqrt_1mac,
                               int32_t b, int32_t n, int32_t p, int32_t c, int32_t num_timesteps, dsc_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Training: hand-written backward of the denoiser (the reference relies on torch autograd through
 * denoise_net.py; train_on_batch, diffusion_scene_layout_ddpm.py:456-473).  Input gradients
 * dA = dY . W reuse dsc_gemm_f32 on transposed weights (dsc_transpose_f32).
 * ------------------------------------------------------------------------------------------- */

/* Weight gradient  out[n][k] = sum_m dy[m][n] * [a1|a2][m][k]  (fp32 MFMA, split over tokens, deterministic
 * two-stage reduction through `workspace`, sized by dsc_gemm_tn_workspace_floats).  Columns >= kvalid of a
 * zero-padded small-K input are not stored; out is dense [n][kvalid] (ldo == kvalid) when split. */
int dsc_gemm_tn_f32(const float* a1, int64_t lda1, int32_t k1, const float* a2, int64_t lda2, int32_t k2,
                    const float* dy, int64_t ldd, float* out, int64_t ldo, float* dbias /* [n] column sums of dy, or NULL */,
                    int32_t m, int32_t n, int32_t kvalid, float* workspace, int64_t workspace_floats, dsc_stream_t stream);
int64_t dsc_gemm_tn_workspace_floats(int32_t m, int32_t n, int32_t k);

/* The same for MANY layers in one launch (static training plan: weight gradients are leaves of the backward pass and every
 * activation stays alive, so they are deferred and computed together -- enough output tiles to fill the chip without cutting
 * the token dimension into 32 slabs per layer).  `groups_dev` is a DEVICE array of `count` descriptors (pointers are device
 * pointers; the constraints of dsc_gemm_tn_f32 apply to each and are validated by the caller that builds the table);
 * tile0 = number of 128 x 128 output tiles of all earlier groups (ceil(n/128) * ceil((k1+k2)/128) each), total_tiles their
 * sum.  splits == 1 writes `out` / `dbias` directly; splits > 1 cuts every group's tokens into `splits` ranges whose partial
 * results go to workspace + ws_offset ([splits][n*kvalid] then [splits][n], per group) and are summed in a fixed order by
 * a second launch.  workspace_needed = end of the last group's area. */
typedef struct dsc_tn_group {
    const float* a1; int64_t lda1; int32_t k1;
    const float* a2; int64_t lda2; int32_t k2;
    const float* dy; int64_t ldd;
    float* out; int64_t ldo;
    float* dbias;
    int32_t m, n, kvalid;
    int32_t tile0;
    int64_t ws_offset;
    int32_t tile0s;   /* reserved (round 3: first tile in the 256 x 128 numbering; the split launch now takes a block map) */
} dsc_tn_group;
int dsc_gemm_tn_grouped_f32(const dsc_tn_group* groups_dev, int32_t count, int32_t total_tiles, int32_t splits,
                            float* workspace, int64_t workspace_floats, int64_t workspace_needed, dsc_stream_t stream);
/* The same launch on the bf16 matrix cores with f32 accuracy (both operands split exactly into three bf16 pieces, six products, f32
 * accumulation; csrc/gemm_tn_split.h).  Output tiles are 256 (n) x 128 (k); the caller supplies the block placement:
 *   block_map_dev  DEVICE array of `blocks` int32 pairs (group index, tile of that group: k tile + ktiles * n tile), 8-byte aligned;
 *                  (-1, -1) = idle block.  Every tile of every group must appear exactly once.  Consecutive workgroup ids are dealt
 *                  round-robin over the 8 XCDs, so entries b, b + 8, b + 16 ... share one L2: put the tiles of one layer there.
 * total_tiles / tile0 stay the 128 x 128 numbering (used by the slab reduction when splits > 1; tile0s is unused).  Every operand must
 * satisfy m * ld * 4 < 2^31.
 * tile_k (round 6): the k width of the tiles the block map numbers -- 128 (256 x 128 tiles: the round-4 / round-5 bodies) or 256 (256 x 256
 * tiles, tile of a group = k tile + ceil(K / 256) * n tile: the round-6 body, which needs k1 % 256 == 0 wherever k2 > 0). */
int dsc_gemm_tn_grouped_split_f32(const dsc_tn_group* groups_dev, int32_t count, int32_t total_tiles, const int32_t* block_map_dev,
                                  int32_t blocks, int32_t splits, float* workspace, int64_t workspace_floats, int64_t workspace_needed,
                                  int32_t tile_k, dsc_stream_t stream);

/* The split-bf16 weight-gradient launch has three block bodies with IDENTICAL results: 2 = 256 x 256 tiles, four waves of 128 x 128 with
 * the 512-register budget (round 6, default: half the staged bytes per MFMA); 1 = 256 x 128 tiles, producer / consumer waves (round 5);
 * 0 = the round-4 body (every wave stages and multiplies).  set returns the previous form (process-wide; launches already captured in a
 * hipGraph keep theirs); DSC_TN_FORM in the environment gives the initial one.  The host asks get() when it builds a block map: form 2
 * -> tile_k 256.  For tests that hold the three to each other bit for bit. */
int dsc_set_tn_split_form(int32_t form);
int dsc_get_tn_split_form(void);

/* out[c] = sum_r x[r][c] (bias / affine gradients); workspace >= 64 * n floats. */
int dsc_colsum_f32(const float* x, int64_t ldx, int32_t m, int32_t n, float* out, float* workspace,
                   int64_t workspace_floats, dsc_stream_t stream);

/* Grouped column sums: out_i[c] = sum_r x_i[r][c] for `count` small matrices in ONE launch (items: a DEVICE array; every m
 * should be a few hundred rows -- the per-scene partials of bias / affine gradients).  Fixed summation order. */
typedef struct dsc_colsum_item { const float* x; int64_t ldx; int32_t m, n; float* out; } dsc_colsum_item;
int dsc_colsum_grouped_f32(const dsc_colsum_item* items_dev, int32_t count, int32_t max_n, dsc_stream_t stream);

/* Backward of the GroupNorm + (scale+1, shift) + SiLU epilogue of dsc_gemm_gn_silu_f32 (Block.forward,
 * denoise_net.py:167-176) from the saved pre-norm output z.  dz feeds the GEMM backward; dgamma_p / dbeta_p /
 * dbias_p are per-scene partials, row b at p + b*partial_stride (reduce with dsc_colsum_f32); dss is [scenes][1024] for
 * DSC_SS_PER_SCENE, [m][1024] for PER_TOKEN / PER_SLOT (caller reduces over the batch for PER_SLOT). */
int dsc_gn_silu_bwd_f32(const float* z, int64_t ldz, const float* dy, int64_t ldy, const float* gamma,
                        const float* beta, const float* scale_shift, int64_t ld_ss, int32_t ss_mode,
                        float* dz, int64_t lddz, float* dgamma_p, float* dbeta_p, float* dbias_p, int64_t partial_stride,
                        float* dss, int64_t ld_dss, int32_t scenes, int32_t tokens_per_scene, int32_t channels,
                        float eps, dsc_stream_t stream);

/* Backward of dsc_weight_standardize_f32: dw = rstd * (dw_std - mean(dw_std) - w_std * mean(dw_std * w_std)). */
typedef struct dsc_ws_bwd_item { const float* w; const float* dw_std; float* dw; int32_t rows; int32_t cols; } dsc_ws_bwd_item;
int dsc_weight_standardize_bwd_f32(const dsc_ws_bwd_item* items, int32_t count, float eps, dsc_stream_t stream);

/* Backward of dsc_layernorm_f32: dx (+ addend when not NULL: the gradient x already holds from another consumer -- dx may be a
 * different buffer than addend, so nothing is modified in place), and per-block partials of the gain gradient [partial_rows][512]. */
int dsc_layernorm_bwd_f32(const float* x, int64_t ldx, const float* g, const float* dy, int64_t ldy, float* dx,
                          int64_t lddx, const float* addend, int64_t ldadd, float* dg_partial, int32_t partial_rows, int32_t m,
                          int32_t d, float eps, dsc_stream_t stream);

int dsc_linear_attention_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                 const float* dout, int64_t ldo, float* dq, int64_t lddq, float* dk, int64_t lddk,
                                 float* dv, int64_t lddv, int32_t scenes, int32_t nq, int32_t nk, float scale,
                                 dsc_stream_t stream);

int dsc_attention_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                          const float* dout, int64_t ldo, float* dq, int64_t lddq, float* dk, int64_t lddk,
                          float* dv, int64_t lddv, int32_t scenes, int32_t n, float scale, dsc_stream_t stream);

/* Training loss of p_losses (diffusion_ddpm.py:556-652, loss_type 'mse', full attribute tensor) and its gradient in
 * one kernel, one block per scene: separated MSE terms x loss_weight[t] + the masked pairwise 3-D IoU regulariser
 * (loss.py:7-102) on the clamped, de-normalised x0 estimate  x0 = ca[t]*x_t - cb[t]*out  (mean_type v / eps; out itself
 * for x0).  losses[b] = losses_weight of scene b; parts[b][9] = {bbox, trans, size, angle, class, object, objfeat,
 * liou, bbox_iou}; dout[b] = grad_scale * d losses[b] / d out[b] (grad_scale = 1/B for loss = losses.mean()).  The
 * re-arrangement model (:558-571; size_dim = class_dim = objectness_dim = objfeat_dim = 0, c = bbox_dim = translation +
 * angle channels) uses losses = l_trans + l_angle, no IoU term.  bounds is a HOST array {centroid min[3], max[3], size min[3],
 * max[3]} (dataset_stats.txt, :137-151), required when loss_iou. */
int dsc_ddpm_loss_f32(const float* target, const float* out, const float* x_t, const int64_t* t,
                      const float* loss_weight, const float* ca, const float* cb, const float* alphas_cumprod,
                      const float* bounds, float* losses, float* parts, float* dout, int32_t b, int32_t n, int32_t c,
                      int32_t translation_dim, int32_t size_dim, int32_t bbox_dim, int32_t class_dim,
                      int32_t objectness_dim, int32_t objfeat_dim, int32_t loss_separate, int32_t loss_iou,
                      int32_t mean_type, float grad_scale, int32_t num_timesteps, dsc_stream_t stream);

/* Strided 2-D copy / accumulate (static training plan: staging of un-aligned column slices, gradient accumulation of
 * multi-consumer activations such as skip connections):  dst[r][c] = src[r][c]   /   dst[r][c] += src[r][c]. */
int dsc_copy2d_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int32_t rows, int32_t cols, dsc_stream_t stream);
int dsc_add2d_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int32_t rows, int32_t cols, dsc_stream_t stream);

/* dx = dy * act'(x) */
int dsc_activation_bwd_f32(const float* x, const float* dy, float* dx, int64_t count, int32_t act, dsc_stream_t stream);

/* out[c][r] = in[r][c] */
int dsc_transpose_f32(const float* in, int64_t ldi, float* out, int64_t ldo, int32_t rows, int32_t cols,
                      dsc_stream_t stream);
/* the same for up to DSC_WS_MAX contiguous matrices in one launch: items[i].out[c][r] = items[i].w[r][c] */
int dsc_transpose_batched_f32(const dsc_ws_item* items, int32_t count, dsc_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Shape retrieval: nearest database object of the same class in the 32-d latent shape-code space,
 * ThreedFutureDataset.get_closest_furniture_to_objfeats (datasets/threed_future_dataset.py:49-59); with sizes
 * (float64, [n][3]) the lexicographic (size mse, feature mse) order of ..._and_size (:61-77, np.lexsort).
 * out_index[q] = database row, -1 when no object carries the label; ties -> lowest index.  Index-exact with the numpy
 * reference (same fp32 summation order).
 * ------------------------------------------------------------------------------------------- */
int dsc_retrieve_nearest_f32(const float* query_feats, const int32_t* query_labels, const double* query_sizes,
                             const float* db_feats, const int32_t* db_labels, const double* db_sizes,
                             int32_t n_query, int32_t n_db, int32_t feat_dim, int32_t* out_index, float* out_dist,
                             dsc_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Training input pipeline: one launch turns B scenes of the HBM-resident cached dataset into the padded
 * (B, N, C) training batch -- RotationAugmentation (threed_front_dataset.py:313-371), Jitter (:559-567),
 * Scale_CosinAngle_ObjfeatsNorm (:481-513), Permutation (:570-584) and the Diffusion padding wrapper (:888-925) fused.
 * The random draws (rotation angle, jitter offsets, object order) are made on the host in the reference's order and
 * passed in; the arithmetic follows numpy's mixed float32/float64 promotion (see oracle/dataset_ref.py).
 *
 * Store (device): offsets[S+1] first object row of each scene; class_labels[total][n_cls_in] one-hot incl. the start
 * and end columns; translations/sizes [total][3]; angles [total]; objfeats [total][feat_dim] or NULL.
 * Batch (device): scene[B]; order[B][N] (object j of the output comes from object order[b][j] of the scene; NULL =
 * stored order); rot[B] radians or NULL (no rotation wrapper); jitter[B][3] (translation, size, angle offsets) or NULL.
 * bounds (HOST, double): {t_lo[3], t_hi[3], s_lo[3], s_hi[3], angle_min, feat_lo, feat_hi}.
 * out[B*N][ld_out] channel order [translation 3 | size 3 | cos, sin | class n_cls_in-1 | objfeat feat_dim]
 * (diffusion_scene_layout_ddpm.py:148-154); rows >= length are the end symbol (class = [-1..-1,+1], rest 0).
 * length[B] = number of objects.  Scenes longer than N are an error (DSC_EINVAL) detected on the host by the caller.
 * ------------------------------------------------------------------------------------------- */
int dsc_encode_scene_batch_f32(const int64_t* offsets, const float* class_labels, const float* translations,
                               const float* sizes, const float* angles, const float* objfeats, int32_t n_cls_in,
                               int32_t feat_dim, const int64_t* scene, const int32_t* order, const double* rot,
                               const double* jitter, int32_t permute_objfeats, const double* bounds, float* out,
                               int64_t ld_out, int64_t* length, int32_t b, int32_t n, dsc_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * 3-D Chamfer distance -- the binding pair of ChamferDistancePytorch/chamfer3D/chamfer_cuda.cpp
 * (chamfer_cuda_forward / chamfer_cuda_backward, kernels chamfer3D.cu:12-171).  xyz1 [b][n][3], xyz2 [b][m][3]
 * contiguous fp32.  dist1[b][n] = squared distance from each point of cloud 1 to its nearest point of cloud 2,
 * idx1 = that point's index (ties -> lowest index); dist2 / idx2 the other way round.  backward writes (does not
 * accumulate into) gradxyz1 / gradxyz2 = d(sum graddist1*dist1 + sum graddist2*dist2)/d xyz; deterministic gather,
 * no atomics.
 * ------------------------------------------------------------------------------------------- */
int dsc_chamfer3d_forward_f32(const float* xyz1, const float* xyz2, float* dist1, float* dist2, int32_t* idx1,
                              int32_t* idx2, int32_t b, int32_t n, int32_t m, dsc_stream_t stream);
int dsc_chamfer3d_backward_f32(const float* xyz1, const float* xyz2, const float* graddist1, const float* graddist2,
                               const int32_t* idx1, const int32_t* idx2, float* gradxyz1, float* gradxyz2, int32_t b,
                               int32_t n, int32_t m, dsc_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Optimizer step of train_on_batch (diffusion_scene_layout_ddpm.py:456-473): torch.nn.utils.clip_grad_norm_ +
 * torch.optim.Adam (networks/__init__.py:29-30) as three launches over a DEVICE work list of chunks (each <= 32768
 * contiguous elements of one parameter; pointers are device pointers).  Nothing is synchronised with the host: the
 * total norm and the clip coefficient stay in device memory and dsc_adam_step_f32 multiplies every gradient by
 * *grad_scale (NULL = 1) while it reads it.  Arithmetic of torch.optim.Adam (amsgrad=False, maximize=False):
 *   g += weight_decay * p;  m += (g - m)(1 - beta1);  v = v beta2 + (1 - beta2) g g;
 *   p -= step_size * m / (sqrt(v) / bias_correction2_sqrt + eps),   step_size = lr / (1 - beta1^t).
 * ------------------------------------------------------------------------------------------- */
typedef struct dsc_optim_chunk {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t count;
} dsc_optim_chunk;

int dsc_grad_sumsq_f32(const dsc_optim_chunk* chunks, int32_t nchunks, double* partial, dsc_stream_t stream);
int dsc_clip_coef_f32(const double* partial, int32_t nchunks, float max_norm, float* total_norm, float* clip_coef,
                      dsc_stream_t stream);
int dsc_adam_step_f32(const dsc_optim_chunk* chunks, int32_t nchunks, float step_size, float beta1, float beta2,
                      float bias_correction2_sqrt, float eps, float weight_decay, const float* grad_scale,
                      dsc_stream_t stream);
/* ---------------------------------------------------------------------------------------------
 * FoldingNet KL auto-encoder (scene_synthesis/networks/foldingnet_autoencoder.py:56-390), the pieces around its 1x1
 * convolutions (which are dsc_gemm_f32 calls).  Point features are token-major [cloud * n + point][channel].
 * ------------------------------------------------------------------------------------------- */
/* knn() (:59-76): the 16 nearest points of every point inside its own cloud (itself included), nearest first; idx [clouds*n][16]
 * holds point indices inside the cloud.  dim == 3 and gram == NULL: distances from the coordinates; otherwise
 * gram = per-cloud Gram matrices [clouds][n][n] (a batched dsc_gemm_f32 of the features with themselves) and sqnorm[rows] =
 * squared row norms (dsc_rowsq_f32).  16 <= n <= 2048. */
int dsc_knn16_f32(const float* x, int64_t ldx, int32_t dim, const float* gram, const float* sqnorm, int32_t clouds, int32_t n,
                  int32_t* idx, dsc_stream_t stream);
int dsc_rowsq_f32(const float* x, int64_t ldx, int32_t dim, int64_t rows, float* out, dsc_stream_t stream);
/* Encoder input (:197-205): out[q][0:3] = xyz, out[q][3:12] = covariance sums of the 16 neighbours (row-major 3x3). */
int dsc_knn_cov_f32(const float* xyz, int64_t ldx, const int32_t* idx, int32_t clouds, int32_t n, float* out, int64_t ldo,
                    dsc_stream_t stream);
/* GraphLayer local max pooling (:160-165) and its backward (dx must be zero-filled; fp32 atomic adds). */
int dsc_gather_max_f32(const float* x, int64_t ldx, const int32_t* idx, int32_t clouds, int32_t n, int32_t ch, float* out,
                       int64_t ldo, uint8_t* arg, dsc_stream_t stream);
int dsc_gather_max_bwd_f32(const float* dy, int64_t ldy, const int32_t* idx, const uint8_t* arg, int32_t clouds, int32_t n,
                           int32_t ch, float* dx_zeroed, int64_t lddx, dsc_stream_t stream);
/* nn.BatchNorm1d over the rows of x [rows][ch] (+ ReLU when relu != 0).  Training form: batch statistics (biased variance for
 * the normalisation, unbiased for running_var; running_* may be NULL), xhat (may be NULL) and y dense [rows][ch];
 * workspace >= dsc_bn_workspace_floats(rows, ch).  Eval form: mean / rstd given.  Backward: dense dy, xhat, y. */
int64_t dsc_bn_workspace_floats(int64_t rows, int32_t ch);
int dsc_batchnorm_fwd_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, int64_t rows, int32_t ch, float eps,
                          float momentum, int32_t relu, float* mean, float* rstd, float* running_mean, float* running_var,
                          float* xhat, float* y, float* workspace, int64_t workspace_floats, dsc_stream_t stream);
int dsc_batchnorm_eval_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, const float* mean,
                           const float* rstd, int64_t rows, int32_t ch, int32_t relu, float* y, dsc_stream_t stream);
int dsc_batchnorm_bwd_f32(const float* dy, const float* xhat, const float* y, const float* gamma, const float* rstd, int64_t rows,
                          int32_t ch, int32_t relu, float* dx, float* dgamma, float* dbeta, float* workspace,
                          int64_t workspace_floats, dsc_stream_t stream);
/* Global max pooling over the points of a cloud (:219): out [clouds][ch], arg = first point attaining it; backward dense. */
int dsc_rowmax_f32(const float* x, int64_t ldx, int32_t clouds, int32_t n, int32_t ch, float* out, int32_t* arg,
                   dsc_stream_t stream);
int dsc_rowmax_bwd_f32(const float* dy, const int32_t* arg, int32_t clouds, int32_t n, int32_t ch, float* dx, dsc_stream_t stream);
/* First convolution of a FoldingLayer (:247-251) on cat([x | codeword]) without the cat:
 * y[b*n + p][c] = sum_{k<d} wp[c][k] * x[row][k] + t[b][c], row = p (x_per_cloud == 0: the shared 2-D grid) or b*n + p, d <= 4,
 * t = Wc . codeword + bias from the GEMM.  Backward: dx (optional), dwp [ch][d], dt [clouds][ch]. */
int dsc_point_affine_f32(const float* x, int64_t ldx, int32_t x_per_cloud, const float* wp, int64_t ldw, const float* t,
                         int32_t clouds, int32_t n, int32_t ch, int32_t d, float* y, dsc_stream_t stream);
int dsc_point_affine_bwd_f32(const float* dy, const float* x, int64_t ldx, int32_t x_per_cloud, const float* wp, int64_t ldw,
                             int32_t clouds, int32_t n, int32_t ch, int32_t d, float* dx, int64_t lddx, float* dwp, int64_t lddw,
                             float* dt, float* workspace, int64_t workspace_floats, dsc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFUSCENE_HIP_H */

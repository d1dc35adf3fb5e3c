/* theatergen_hip.h — C ABI of libtheatergen_hip.so (hand-written HIP for gfx950 / MI355X).
 *
 * The reference (donahowe/TheaterGen) is 100 % Python on PyTorch and has NO FFI; this boundary is new
 * (last row).  Each entry point replaces the eager PyTorch op sequence of one piece of
 * the per-character denoising hot path; the reference lines it replaces are cited on each declaration
 * (paths relative to the reference repo).  The reference-side binding a maintainer would add is the
 * ctypes stub shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers (tensor.data_ptr()), sizes, POD descriptors in HOST memory;
 *     no torch / C++ types.  Caller owns every buffer; outputs are caller-allocated; no ownership
 *     transfer; nothing is retained after return.
 *   - `stream` is a hipStream_t passed as void*; every call is stream-ordered, asynchronous and
 *     re-entrant (no global mutable state except the thread-local error string).
 *   - return 0 on success, negative TG_ERR_* otherwise; tg_last_error() gives the message.  The Python
 *     shim maps any non-zero code to RuntimeError so the caller policy of reference generate.py:250-259
 *     ("RuntimeError => skip turn") is preserved.
 *   - dtype: TG_BF16 / TG_F16 select the storage type of activations and weights; all accumulation,
 *     softmax and normalisation statistics are fp32.
 *   - activations are token-major ("NHWC"): [batch, h*w, channels], channels contiguous.
 */
#ifndef THEATERGEN_HIP_H
#define THEATERGEN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TG_BF16 0
#define TG_F16 1

#define TG_OK 0
#define TG_ERR_ARG (-1)      /* invalid / unsupported argument */
#define TG_ERR_LAUNCH (-2)   /* HIP launch error */
#define TG_ERR_UNSUPPORTED (-3)

#define TG_ACT_NONE 0
#define TG_ACT_SILU 1
#define TG_ACT_GELU 2        /* exact (erf) GELU */
#define TG_ACT_QUICK_GELU 3  /* x * sigmoid(1.702 x): OpenAI CLIP's activation (text / image encoders) */

/* ABI version: changes whenever a signature or descriptor layout in this header changes.  Callers compare it with the
 * TG_ABI_VERSION they were built against (the ctypes binding does at load time) and refuse a mismatching library. */
#define TG_ABI_VERSION 307
int tg_version(void);
const char* tg_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * GEMM / implicit-GEMM 3x3 convolution on MFMA:   out[m, n] = epilogue( sum_k A[m, k] * W[n, k] )
 * Replaces nn.Linear / nn.Conv2d of:
 *   Attention.to_q/to_k/to_v/to_out      ip_adapter/attention_processor.py:113-128, 333-362
 *   IPAttnProcessor.to_k_ip/to_v_ip      ip_adapter/attention_processor.py:418-419, 497-498
 *   Transformer2DModel.proj_in/proj_out  models/transformer_2d.py:286-293, 322-328
 *   FeedForward / GEGLU                  models/attention.py:243-292, 317-338
 *   ResnetBlock2D conv1/conv2/shortcut/time_emb_proj, Downsample2D, Upsample2D
 *                                        (diffusers 0.21.4; call sites models/unet_2d_blocks.py:184-195,
 *                                         313-324, 355-362, 477-488, 578-589, 620-622, 742-753)
 *   TimestepEmbedding                    models/unet_2d_condition.py:315-328
 *   Resampler linears                    ip_adapter/resampler.py:13-20, 45-47, 94-97
 * mode 0: A is row-major [M, K] (K = c0 + c1 when a1 != NULL: channel-concat of two sources with
 *         row pitches c0 / c1 — the skip-connection cat of models/unet_2d_blocks.py:648-651 without a copy).
 * mode 1: A is gathered on the fly from NHWC activations [batch, in_h, in_w, c0 (+c1)] for a 3x3,
 *         pad-1 convolution with `stride` 1|2, optionally on the nearest-neighbour x2 upsampled input
 *         (`upsample`=1, Upsample2D); W is [N, 9*(c0+c1)] tap-major (ky, kx, c).  M = batch*out_h*out_w.
 * Epilogue (all optional, fp32):  v = acc + bias[n] + bvec[m / rows_per_batch, n] + res[m, n];
 *         v = act(v) * out_scale;  act in TG_ACT_*;  or GEGLU (geglu=1, bias only): W rows (and bias) are PACKED
 *         per 64-row group as [a(32) ; gate(32)] (theatergen_amd.weights_pack.pack_geglu of GEGLU.proj), N % 64 == 0,
 *         out is [M, N/2] = (a + b_a) * gelu(gate + b_g) * out_scale: the [M, N] pre-activation never reaches HBM.
 * Output: out[m * ldc + n] for n < n_split (n_split <= 0: all); columns n >= n_split are written
 *         TRANSPOSED per batch item, out_t[(b * (N - n_split) + n - n_split) * ldt + (m % rows_per_batch)]
 *         (the V^T operand of tg_attention).
 * `workspace`: fp32 scratch of `workspace_bytes`, used when the kernel splits K (small-M layers);
 *         tg_gemm_workspace_bytes() gives the size needed for a descriptor.
 */
typedef struct {
  int32_t dtype;
  int32_t mode;
  const void* a0;
  const void* a1;
  int32_t c0, c1;
  int32_t batch, in_h, in_w, out_h, out_w;
  int32_t stride, upsample;
  const void* w;
  int64_t M, N, K;
  const void* bias;      /* [N] dtype, or NULL */
  const void* bvec;      /* [M / rows_per_batch, N] dtype, or NULL */
  int64_t ldbvec;        /* row pitch of bvec (elements) */
  int64_t rows_per_batch;
  const void* res;       /* [M, ldres] dtype, or NULL */
  int64_t ldres;
  int32_t act;
  int32_t geglu;
  float out_scale;
  void* out;
  int64_t ldc;
  int64_t n_split;
  void* out_t;
  int64_t ldt;
  void* workspace;
  int64_t workspace_bytes;
  int32_t force_split_k; /* 0 = heuristic; >0 forces that many K splits (testing) */
  int32_t force_tile;    /* 0 = heuristic; otherwise tile config id (testing) */
  /* mode 0 only: A rows grouped per batch item with a batch pitch (elements): row m lives at
   * a0 + (m / a_rows_per_batch) * a_batch_stride + (m % a_rows_per_batch) * c0.  0 = contiguous.
   * (slices encoder_hidden_states[:, :L-T] / [:, L-T:] without a copy, attention_processor.py:467-471) */
  int64_t a_rows_per_batch;
  int64_t a_batch_stride;
  /* mode 1 only: 0 = symmetric zero padding 1 (every UNet conv); 1 = padding on the bottom / right edge only, the
   * `F.pad(x, (0, 1, 0, 1))` + stride-2 conv of diffusers' Downsample2D(padding=0) in the VAE ENCODER
   * (AutoencoderKL.encode, models/pipelines.py:157, 624): input pixel = stride * out + tap, no -1 offset */
  int32_t pad_mode;
  /* mode 1, stride 1, N % 320 == 0 only (the slab kernel, tg_gemm_plan kernel_kind 4): GroupNorm (+ SiLU) of the INPUT applied
   * while the window is staged: A'[b, p, c] = act(A[b, p, c] * a[b, c] + d[b, c]) with a = rstd * gamma, d = beta - mean * a from
   * tg_groupnorm_coef ([batch][2][c0 + c1] fp32: a then d per batch item); zero padding stays zero.  Same fp32 expression and
   * storage-dtype rounding as tg_groupnorm, so conv(A', W) is bit-identical to tg_groupnorm followed by the plain conv; the
   * normalised tensor never exists in HBM (ResnetBlock2D: conv1(nonlinearity(norm1(x))), conv2(nonlinearity(norm2(h))),
   * models/unet_2d_blocks.py:184-195 via diffusers ResnetBlock2D.forward).  NULL = plain conv.  TG_ERR_UNSUPPORTED when the
   * problem is not one the slab kernel takes (ask tg_gemm_plan first). */
  int32_t a_silu;        /* 1: act = SiLU, 0: identity */
  const void* a_coef;
  /* mode 0, one A source, K = the LayerNorm width, linear or GEGLU epilogue without residual / per-batch vector (tg_gemm_plan
   * kernel_kind 6): LayerNorm(K, ln_eps) of every A ROW folded into the GEMM — BasicTransformerBlock's norm1 -> attn1 q|k|v,
   * norm2 -> attn2.to_q, norm3 -> ff.net.0.proj (models/attention.py:186-236) without the normalised tensor or a layernorm launch:
   *     out[m, n] = epilogue( rstd[m] * (sum_k A[m, k] W'[n, k] - mean[m] * ln_u[n]) + ln_v[n] )
   * with W' = W * gamma (the caller packs it in the storage dtype and passes it as `w`), ln_u[n] = sum_k W'[n, k] over the PACKED
   * (rounded) values, ln_v[n] = sum_k beta[k] W[n, k] + bias[n]; fp32 [N], 16-byte aligned; `bias` must be NULL (it lives in
   * ln_v).  mean / rstd are taken inside the kernel from the A tiles it stages (biased variance over the stored values, as
   * nn.LayerNorm).  NULL = plain GEMM. */
  const float* ln_u;
  const float* ln_v;
  float ln_eps;
  const float* ln_rows;  /* optional with ln_u: precomputed row statistics (tg_layernorm_stats, fp32 [M][2]); NULL = taken inside the kernel */
  /* round 5, mode 0 with one A source only (plain GEMM kernels; 0 = K): row pitches of A and W in elements, multiples of 8, >= K.  A row pitch that is a large
   * power-of-two multiple (K = 2560 / 5120: the FeedForward hidden tensor and net.2's weight) puts the rows of a K-tile on few L2 channels when the workgroups of a
   * single-round launch run in lockstep (profiles/r5_operand_pitch.txt: 594 -> 718 TFLOP/s at K = 5120 with the pitch padded by 64 elements); the caller
   * (unet.FeedForward) pads the hidden tensor it owns and its packed copy of net.2's weight. */
  int64_t lda;
  int64_t ldw;
  /* round 6, mode 1 on the two-wave slab kernel without a K split only (tg_gemm_gn_partial_blocks(d) > 0): the GroupNorm(out_gn_groups) partial sums of the
   * OUTPUT leave the epilogue, so the consumer's statistics pass (gn_partial_kernel: one more HBM read of the tensor) disappears: every compute wave sums the
   * stored (rounded) values of its 64 pixels x 80 channels per group — fp32 [batch][hw / 64][out_gn_groups][2] = (sum, sum of squares), each entry written by
   * exactly one wave in a fixed order (deterministic) — the layout tg_groupnorm_from_partials folds (fp64) into the statistics.  ResnetBlock2D: conv1 -> norm2,
   * conv2 (+ shortcut) -> Transformer2DModel.norm (models/unet_2d_blocks.py:184-195, models/transformer_2d.py:303-316 of diffusers).  NULL = off;
   * TG_ERR_UNSUPPORTED when the planner's kernel for the descriptor does not write them. */
  float* out_gn_partials;
  int32_t out_gn_groups;
} tg_gemm_desc;

int tg_gemm(const tg_gemm_desc* d, void* stream);
int64_t tg_gemm_workspace_bytes(const tg_gemm_desc* d);
/* the tile (tokens x channels), the K-split of the tail tiles (1 = none) and the kernel (0 = GEMM, 1 = implicit-GEMM
 * conv, 2 = LDS-halo conv, 3 = big-tile persistent GEMM, 4 = slab conv (128 x 320 tiles, loader + compute waves, optional
 * GroupNorm prologue), 5 = loader / compute GEMM (128 x 320 tiles, long K), 6 = LayerNorm-fused projection (ln_u)) the heuristic
 * picks for a descriptor */
int tg_gemm_plan(const tg_gemm_desc* d, int32_t* tile_m, int32_t* tile_n, int32_t* splits, int32_t* kernel_kind);
/* > 0: the kernel tg_gemm runs for `d` can write GroupNorm(d->out_gn_groups) partial sums of its output (tg_gemm_desc.out_gn_partials); the value is the number
 * of 64-pixel blocks per batch item (the `nblk` of tg_groupnorm_from_partials).  0: it cannot (ask before setting out_gn_partials). */
int tg_gemm_gn_partial_blocks(const tg_gemm_desc* d);

/* ---------------------------------------------------------------------------------------------
 * Fused flash-style attention with up to two independently-normalised K/V segments:
 *     O = softmax(s Q K0^T) V0  +  w1 * softmax(s Q K1^T) V1
 * Segment 0 alone = AttnProcessor self/cross attention (ip_adapter/attention_processor.py:187-219,
 * 346-357: baddbmm -> softmax -> bmm, here without materialising [B*h, N, Lk]); both segments =
 * IPAttnProcessor's decoupled text + image cross-attention (:475-516, TWO softmaxes, w1 = scale);
 * also PerceiverAttention (ip_adapter/resampler.py:69-75; s = d^-0.5 = (d^-0.25)^2).
 * Layouts (elements of dtype): q[b, i, h*d + c] with row pitch q_ld and batch pitch q_bs;
 * k{0,1}[b, j, h*d + c] likewise; vt{0,1} is V TRANSPOSED per batch: vt[b, h*d + c, j], row pitch vt_ld
 * (multiple of 8), batch pitch vt_bs.  out[b, i, h*d + c].  head_dim in {40, 64, 80, 160} (or any
 * multiple of 8 <= 160).  len1 == 0 disables segment 1.
 */
typedef struct {
  int32_t dtype;
  int32_t batch, heads, head_dim;
  int32_t n_q;
  const void* q; int64_t q_ld, q_bs;
  const void* k0; int64_t k0_ld, k0_bs;
  const void* vt0; int64_t vt0_ld, vt0_bs;
  int32_t len0;
  const void* k1; int64_t k1_ld, k1_bs;
  const void* vt1; int64_t vt1_ld, vt1_bs;
  int32_t len1;
  float scale;
  float w1;
  void* out; int64_t out_ld, out_bs;
  int32_t causal;   /* 1: key j is visible to query i only if j <= i (segment 0; the CLIP text encoder's mask) */
  const float* w1_dev;  /* non-NULL: segment 1's weight is READ FROM THE DEVICE at run time (one fp32) and `w1` is ignored — the IP scale that
                         * IPAdapter.set_scale mutates per character and ip_adapter/custom_pipelines.py:328-333 gates per step: a captured
                         * hipGraph of the step replays with the current value */
  const float* mask;    /* non-NULL: additive bias on segment 0's scores, fp32, score units (the diffusers `attention_mask` after
                         * `prepare_attention_mask`, ip_adapter/attention_processor.py:193-206, 221-259):
                         * mask[b * mask_bs + h * mask_hs + i * mask_qs + j]; zero strides broadcast over heads / queries */
  int64_t mask_bs, mask_hs, mask_qs;
} tg_attn_desc;

int tg_attention(const tg_attn_desc* d, void* stream);

/* Reverse pass of SELF-attention without materialised probabilities (round 5, ABI 306; csrc/tg_attention_bwd.hip), head_dim <= 64, n % 8 == 0:
 *     dQ = dS K,  dK = dS^T Q,  dV = P^T dO   with  P = softmax(scale Q K^T),  dS = scale P o (dO V^T - rowsum(P o dO V^T))
 * — what `torch.autograd.grad(loss, latents)` of the guidance step (models/pipelines.py:62-128) computes through `Attention` / AttnProcessor
 * (ip_adapter/attention_processor.py:113-219).  Three launches of one kernel (row statistics lse / D, dQ, dK + dV); scores are recomputed from Q / K
 * tiles, nothing n x n exists in memory.  q, k, v, dout, dq, dk, dv: [batch][n][>= heads * head_dim] rows of pitch `ld`, batch stride `bs` (elements);
 * qt, kt, doutt: the TRANSPOSES [batch][heads * head_dim][n] (tg_transpose) with row pitch `t_ld`, batch stride `t_bs`; stats: fp32 scratch
 * [batch][heads][n][2].  P and dS are rounded to the storage dtype where they enter a product (as a materialised implementation stores them).
 * TG_ERR_UNSUPPORTED for other head dims / ragged n: the caller keeps its materialised path. */
typedef struct {
  int32_t dtype, batch, heads, head_dim, n;
  const void* q; const void* k; const void* v; const void* dout;
  int64_t ld, bs;
  const void* qt; const void* kt; const void* doutt;
  int64_t t_ld, t_bs;
  float* stats;
  void* dq; void* dk; void* dv;
  float scale;
} tg_attn_bwd_desc;
int tg_attention_bwd(const tg_attn_bwd_desc* d, void* stream);
/* The same for one softmax segment of CROSS-attention (text keys, or the IP-Adapter's image keys: ip_adapter/attention_processor.py:445-529): K / V are
 * constants of the conditioning, so only dQ is produced (statistics + dQ launches), over n_k keys (any count; `kt` = K^T [batch][heads * head_dim][t_ld]
 * zero-padded to a multiple of 8 columns).  `extra` (optional, fp32 [batch][heads][n_q][extra_ld >= n_k]): d loss / d P of a loss that reads the
 * probabilities (the guidance loss on the text maps, utils/guidance.py:91-286) — it joins dO V^T before the softmax Jacobian.  `scale`: softmax scale;
 * `ds_scale`: scale x the segment's output weight (the IP scale for the image segment).  dq is written (not accumulated). */
typedef struct {
  int32_t dtype, batch, heads, head_dim, n_q, n_k;
  const void* q; const void* dout; int64_t q_ld, q_bs;
  const void* k; const void* v; int64_t k_ld, k_bs;
  const void* kt; int64_t t_ld, t_bs;
  const float* extra; int64_t extra_ld;
  float* stats;                /* fp32 scratch [batch][heads][n_q][2] */
  void* dq;
  float scale, ds_scale;
} tg_attn_bwd_cross_desc;
int tg_attention_bwd_cross(const tg_attn_bwd_cross_desc* d, void* stream);

/* Attention-probability export (the save_attn_to_dict side channel, attention_processor.py:532-551):
 * probs[b - b0, h, i, t] = softmax_j(s q_i . k_j)[tokens[t]] for batch items b in [b0, batch), fp32 out
 * [batch - b0, heads, n_q, n_tokens].  tokens == NULL: all `len` columns.  */
int tg_attn_probs(int32_t dtype, int32_t batch, int32_t b0, int32_t heads, int32_t head_dim, int32_t n_q,
                  const void* q, int64_t q_ld, int64_t q_bs, const void* k, int64_t k_ld, int64_t k_bs,
                  int32_t len, float scale, const int32_t* tokens, int32_t n_tokens, float* probs, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Normalisation (HBM-bound).
 * tg_groupnorm: GroupNorm(groups, eps) over NHWC [batch, hw, c0(+c1)] (+ optional SiLU) -> out [batch*hw, C].
 *   Two sources = channel concat (skip connection).  `partials` fp32 scratch of
 *   tg_groupnorm_scratch_bytes(batch, hw, groups).  Replaces ResnetBlock2D.norm1/norm2 + nonlinearity,
 *   Transformer2DModel.norm (models/transformer_2d.py:146, 285), conv_norm_out + conv_act
 *   (models/unet_2d_condition.py:1015-1017).
 * tg_layernorm: LayerNorm(C, eps) per row: BasicTransformerBlock.norm1/2/3 (models/attention.py:186, 206, 226),
 *   Resampler norms (ip_adapter/resampler.py:43-44, 66-67, 100).  gamma/beta may be NULL.
 */
int64_t tg_groupnorm_scratch_bytes(int32_t batch, int64_t hw, int32_t groups);
int tg_groupnorm(int32_t dtype, const void* x0, const void* x1, int32_t c0, int32_t c1, int32_t batch, int64_t hw,
                 int32_t groups, float eps, const void* gamma, const void* beta, int32_t silu, void* out,
                 void* partials, void* stream);
int tg_layernorm(int32_t dtype, const void* x, int64_t rows, int32_t C, int64_t ldx, float eps, const void* gamma,
                 const void* beta, void* out, int64_t ldo, void* stream);
/* Row statistics of tg_layernorm only: stats[m] = (rstd, -rstd * mean) fp32 [rows][2] (two-pass mean / centred biased variance over the
 * stored values) — the `ln_rows` input of the LayerNorm-folded projections (tg_gemm_desc.ln_u): half the traffic of the normalising pass. */
int tg_layernorm_stats(int32_t dtype, const void* x, int64_t rows, int32_t C, int64_t ldx, float eps, float* stats, void* stream);
/* GroupNorm statistics only: coef[b][0][c] = rstd(b, group(c)) * gamma[c], coef[b][1][c] = beta[c] - mean(b, group(c)) * coef[b][0][c]
 * (fp32 [batch][2][c0 + c1]; the same reductions, in the same order, as tg_groupnorm) for tg_gemm_desc.a_coef: the apply pass
 * (normalise + SiLU, one HBM write + read of the activation per conv) moves into the consumer conv's window staging. */
int tg_groupnorm_coef(int32_t dtype, const void* x0, const void* x1, int32_t c0, int32_t c1, int32_t batch, int64_t hw,
                      int32_t groups, float eps, const void* gamma, const void* beta, float* coef, void* partials, void* stream);

/* tg_groupnorm / tg_groupnorm_coef from partial sums a producer already wrote (tg_gemm_desc.out_gn_partials: fp32 [batch][nblk][groups][2], one entry per
 * 64-pixel block): the statistics launch is skipped.  `coef` != NULL: the coefficients (x is not read); else `out` = the normalised (+ SiLU) tensor of the
 * single-source x [batch, hw, C].  The fold of the partials (fp64, fixed order) and the apply expressions are tg_groupnorm's; the sums were accumulated in a
 * different order than gn_partial_kernel's, so results agree with tg_groupnorm to rounding of the statistics, not bit for bit. */
int tg_groupnorm_from_partials(int32_t dtype, const void* x, int32_t C, int32_t batch, int64_t hw, int32_t groups, float eps, const void* gamma,
                               const void* beta, int32_t silu, void* out, float* coef, const float* partials, int32_t nblk, void* stream);

/* out[m, j] = x[m, j] * gelu(x[m, inner + j])  (GEGLU.forward, models/attention.py:337-338) */
int tg_geglu(int32_t dtype, const void* x, int64_t rows, int64_t inner, void* out, void* stream);
/* out = act(x) elementwise (n elements) */
int tg_act(int32_t dtype, const void* x, int64_t n, int32_t act, void* out, void* stream);
/* out = a + b (n elements; ControlNet residual injection, models/unet_2d_condition.py:938-946, 975-976) */
int tg_add(int32_t dtype, const void* a, const void* b, int64_t n, void* out, void* stream);

/* dst[b, c, r] = src[b, r, c]: NCHW <-> token-major conversion for the processors' 4-D input path
 * (attention_processor.py:318-320, 363-364) and ControlNet residual injection (models/unet_2d_condition.py:938-946). */
int tg_transpose(int32_t dtype, const void* src, int32_t batch, int32_t rows, int32_t cols, void* dst, void* stream);

/* VAE decode helpers (AutoencoderKL.decode as called at models/pipelines.py:468, 849-854; diffusers 0.21.4):
 * tg_conv1x1_nchw : out[b, o, p] = bias[o] + sum_c w[o, c] * (in_scale * x[b, c, p]), fp32 NCHW, cin / cout <= 8
 *                   (post_quant_conv with the `latents / vae.config.scaling_factor` division folded in)
 * tg_softmax_rows : out[r, :] = softmax(scale * x[r, :]) per row (fp32 math), the single-head d = 512 attention of the
 *                   VAE mid block is scores-GEMM -> this -> PV-GEMM (head dims > 160 are outside tg_attention) */
int tg_conv1x1_nchw(const float* x, int32_t batch, int32_t cin, int32_t cout, int64_t hw, const float* weight,
                    const float* bias, float in_scale, float* out, void* stream);
int tg_softmax_rows(int32_t dtype, const void* x, int64_t rows, int32_t cols, int64_t ldx, float scale, void* out,
                    int64_t ldo, void* stream);

/* ---------------------------------------------------------------------------------------------
 * UNet boundary convolutions (tiny channel counts, direct):
 * tg_conv_in : sample NCHW [batch, cin, h, w] (src_dtype: 0 bf16, 1 f16, 2 f32) -> NHWC [batch, h*w, cout],
 *              3x3 pad 1, weight [cout, 3, 3, cin] + bias  (conv_in, models/unet_2d_condition.py:294-297, 879)
 * tg_conv_out: NHWC [batch, h*w, cin] (already GroupNorm+SiLU'ed) -> NCHW fp32|dtype [batch, cout, h, w]
 *              (conv_out, models/unet_2d_condition.py:583-586, 1018)
 */
int tg_conv_in(int32_t dtype, const void* sample, int32_t src_dtype, int32_t batch, int32_t cin, int32_t h, int32_t w,
               const void* weight, const void* bias, int32_t cout, void* out, void* stream);
int tg_conv_out(int32_t dtype, const void* x, int32_t batch, int32_t cin, int32_t h, int32_t w, const void* weight,
                const void* bias, int32_t cout, void* out, int32_t out_f32, void* stream);
/* round 5 (ABI 306): conv_norm_out + SiLU + conv_out in one launch (models/unet_2d_condition.py:1015-1018): x is the RAW block output, `coef` the
 * tg_groupnorm_coef coefficients of conv_norm_out (fp32 [batch][2][cin]); A'[b, p, c] = act(x[b, p, c] * a[b, c] + d[b, c]) is formed while the tile's
 * window is staged (same fp32 expression and rounding as tg_groupnorm: the normalised tensor never exists in HBM).  Only for problems the matrix-core
 * kernel takes — tg_conv_out_takes_coef(cin, h, w, cout) == 1 (cin % 64 == 0, cin <= 320, cout <= 8, h % 8 == 0, w % 16 == 0) — else TG_ERR_UNSUPPORTED. */
int tg_conv_out_gn(int32_t dtype, const void* x, const float* coef, int32_t a_silu, int32_t batch, int32_t cin, int32_t h, int32_t w,
                   const void* weight, const void* bias, int32_t cout, void* out, int32_t out_f32, void* stream);
int tg_conv_out_takes_coef(int32_t cin, int32_t h, int32_t w, int32_t cout);

/* Sinusoidal timestep embedding (diffusers Timesteps; call site models/unet_2d_condition.py:315-316, 819):
 * out[r, :] for r < rows; t read from DEVICE fp32 array `t` (stride t_stride, 0 = broadcast one value).  When
 * `index` (DEVICE int32, may be NULL) is given the value is t[*index + r * t_stride]: the whole 50-step schedule
 * lives on the device and a captured hipGraph of one step replays without host updates. */
int tg_timestep_embedding(int32_t dtype, const float* t, const int32_t* index, int32_t t_stride, int32_t rows, int32_t dim,
                          int32_t flip_sin_to_cos, float freq_shift, void* out, int64_t ldo, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Step epilogue: CFG combine + DDIM update (+ frozen-mask replace) in one pass over the latents
 * (models/pipelines.py:441-447 and 833-834).  noise_pred: [2*n_img, C, h, w] fp32 (uncond half first) when
 * has_cfg, else [n_img, C, h, w] used as the model output directly (plain scheduler.step);
 * latents fp32 [n_img, C, h, w] updated IN PLACE.  coef: DEVICE fp32 table [n_steps][4] =
 * {sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)}; step_idx: DEVICE int32 (read, then
 * incremented when `advance`).  prediction_type 0 = epsilon, 1 = v_prediction.
 * frozen (may be NULL): fp32 [n_steps+1, n_img, C, h, w]; the row step_idx+1 is blended in with
 * frozen_mask fp32 [n_img|1, h, w] while step_idx < frozen_steps.  history (may be NULL):
 * fp32 [n_steps+1, n_img, C, h, w], row step_idx+1 receives the new latents (latents_all, :449-453, 488).
 */
int tg_step_epilogue(const float* noise_pred, float* latents, int32_t n_img, int32_t chw, int32_t hw,
                     int32_t has_cfg, float guidance_scale, const float* coef, int32_t* step_idx, int32_t advance,
                     int32_t prediction_type, const float* frozen, const float* frozen_mask, int32_t mask_per_img,
                     int32_t frozen_steps, float* history, void* model_in, int32_t model_in_dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Input-gradient kernels of `latent_backward_guidance` (models/pipelines.py:62-128: autograd.grad(loss, [latents]) through the
 * UNet).  Weights are frozen, so only d loss / d input exists; the contractions of the backward pass are tg_gemm on transposed /
 * tap-flipped weights, these are the remaining Jacobians (fp32 arithmetic, storage dtype in / out, deterministic):
 * tg_groupnorm_bwd  : GroupNorm(+SiLU) wrt its token-major input x [batch*hw, C]       (ResnetBlock2D / Transformer2DModel norms)
 * tg_layernorm_bwd  : LayerNorm wrt its input [rows, C]                                 (models/attention.py:186-236)
 * tg_geglu_bwd      : h = [a | gate] [rows, 2*inner], dg [rows, inner] -> dh            (models/attention.py:337-338)
 * tg_softmax_bwd_rows: dS = scale * P * (dP + extra - sum_j P_j (dP_j + extra_j)) per row; P fp32 (tg_attn_probs, probs_fp32 = 1) or
 *                     in the storage dtype (tg_softmax_rows of a scores GEMM: self-attention with more than 256 keys), dP storage dtype,
 *                     extra (may be NULL) fp32 = d loss / d P from tg_guidance_*; dS (and optionally P) written in the storage
 *                     dtype with pitch ld_out >= length, pad columns zeroed            (ip_adapter/attention_processor.py:187-219)
 * tg_sumpool2x2     : sum over the 2 x 2 block of an upsampled token-major map [batch, 2h, 2w, C] -> [batch, h, w, C] (Upsample2D)
 */
/* `partials`: fp32 scratch of tg_groupnorm_bwd_scratch_bytes(batch, hw, groups) — the rows of a batch item are cut into slabs so that the
 * batch-1 backward pass spreads over the chip (three launches: statistics, gradient sums, apply; fixed-order folds).  NULL, or channels not a
 * multiple of 8, runs the one-workgroup-per-(item, group) kernel. */
int64_t tg_groupnorm_bwd_scratch_bytes(int32_t batch, int64_t hw, int32_t groups);
int tg_groupnorm_bwd(int32_t dtype, const void* x, const void* dy, int32_t batch, int64_t hw, int32_t channels, int32_t groups, float eps,
                     const void* gamma, const void* beta, int32_t silu, void* dx, void* partials, void* stream);
int tg_layernorm_bwd(int32_t dtype, const void* x, const void* dy, int64_t rows, int32_t channels, float eps, const void* gamma, void* dx,
                     void* stream);
int tg_geglu_bwd(int32_t dtype, const void* h, const void* dg, int64_t rows, int64_t inner, void* dh, void* stream);
int tg_softmax_bwd_rows(int32_t dtype, const void* probs, int32_t probs_fp32, int64_t ld_probs, const void* dprobs, int64_t ld_dprobs,
                        const float* extra, int64_t ld_extra, int64_t rows, int32_t length, float scale, void* dscores, void* probs_out,
                        int64_t ld_out, void* stream);
int tg_sumpool2x2(int32_t dtype, const void* du, int32_t batch, int32_t h, int32_t w, int32_t channels, void* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Latent utilities (utils/latents.py, utils/utils.py) on fp32 latents [.., h, w]:
 * tg_blend_latents : bg (1-M) + (bg sqrt(1-r) + fg sqrt(r)) M                      (latents.py:156-166)
 *                     storage_dtype -1: fp32 latents; TG_F16 / TG_BF16: the inputs hold half-precision values (the
 *                     reference draws and blends in unet.dtype, latents.py:261-288) and every half-precision tensor op
 *                     of the reference expression rounds to that type -> output exactly representable in it
 * tg_shift         : zero-filled integer shift of the last two dims               (utils.py:143-178)
 * tg_masked_compose: dst = dst (1-M) + src M over `planes` planes of h*w           (latents.py:203-214)
 */
int tg_blend_latents(const float* bg, const float* fg, const float* mask, int32_t planes, int32_t hw, float ratio,
                     float sigma, int32_t storage_dtype, float* out, void* stream);
int tg_shift(const float* src, int64_t planes, int32_t h, int32_t w, int32_t dx, int32_t dy, float* dst, void* stream);
/* DiagonalGaussianDistribution.sample of AutoencoderKL.encode (diffusers 0.21.4; reference models/pipelines.py:157, 624-626):
 * moments fp32 [B, 2C, hw] = (mean | logvar); out[b, c, i] = scale * (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise[b, c, i]);
 * noise == NULL gives the mode (scale * mean).  `scale` = vae.config.scaling_factor (:159, :626). */
int tg_gaussian_sample(const float* moments, const float* noise, int32_t batch, int32_t channels, int32_t hw, float scale,
                       float* out, void* stream);
/* scheduler.add_noise over a table of timesteps (models/pipelines.py:629-631): out[s, i] = ca[s] * x0[i] + cb[s] * noise[i],
 * x0 / noise fp32 [n], ca / cb fp32 [steps] (sqrt(alpha_bar_t), sqrt(1 - alpha_bar_t)), out fp32 [steps, n] */
int tg_add_noise(const float* x0, const float* noise, const float* ca, const float* cb, int32_t steps, int64_t n, float* out,
                 void* stream);
int tg_masked_compose(float* dst, const float* src, const float* mask, int64_t planes, int32_t hw, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-box guidance reductions (utils/guidance.py:91-148, 223-233) on an attention map
 * attn fp32 [heads, hw, n_tok]; mask fp32 [hw] (box mask).  One (token) column per call-row:
 * tg_guidance_topk: for each head: fg = mean(topk_{k_fg}(A*M)), bg = mean(topk_{k_bg}(A*(1-M)));
 *    out[0] += fg_w * sum_h (1 - fg) + bg_w * sum_h bg   (scaled by `scale`)
 *    grad (may be NULL) [heads, hw, n_tok] += d out / d A.
 * tg_guidance_ratio: out[0] += scale * mean_h (1 - sum(A*M)/sum(A))^2  (+ grad)
 * tg_guidance_ref (attention transfer, utils/guidance.py:150-242, inner term :223-233): ref fp32 [heads, hw] (the saved
 *    reference column ``ref_ca_saved_attns[obj][index][key][0, :, :, 0]``):
 *    out[0] += scale * mean_h sum_i | A_i M_i / (sum(A M) + eps) - R_i M_i / (sum(R M) + eps) |   (+ grad wrt A)
 */
int tg_guidance_topk(const float* attn, int32_t heads, int32_t hw, int32_t n_tok, int32_t token, const float* mask,
                     int32_t k_fg, int32_t k_bg, float fg_w, float bg_w, float scale, float* out, float* grad,
                     void* stream);
int tg_guidance_ratio(const float* attn, int32_t heads, int32_t hw, int32_t n_tok, int32_t token, const float* mask,
                      float scale, float* out, float* grad, void* stream);
int tg_guidance_ref(const float* attn, int32_t heads, int32_t hw, int32_t n_tok, int32_t token, const float* ref,
                    const float* mask, float eps, float scale, float* out, float* grad, void* stream);
/* One launch for a whole compute_ca_lossv3 call (utils/guidance.py:244-286: 4 keys x objects x token positions): block i
 * evaluates items[i] (kind 0 = top-k term :130-144, 1 = ratio term :122-128, 2 = reference-attention term :223-233, fields
 * as in the three entry points above) into partials[i]; a second kernel then adds the partials to out[0] IN ITEM ORDER, the
 * same sequence of additions the per-item launches perform (bit-identical loss).  `items_device` lives in device memory;
 * items of one call must not share a (grad, token) column; max_hw_topk = largest hw among the kind-0 items (0 if none),
 * max_heads = largest head count: they size the dynamic LDS (TG_ERR_ARG when a map is too large for the select). */
typedef struct {
  const float* attn;
  float* grad;          /* or NULL */
  const float* mask;
  const float* ref;     /* kind 2 only */
  int32_t heads, hw, n_tok, token;
  int32_t kind, k_fg, k_bg, reserved;
  float fg_w, bg_w, scale, eps;
} tg_guidance_item;
int tg_guidance_batch(const tg_guidance_item* items_device, int32_t n_items, int32_t max_hw_topk, int32_t max_heads,
                      float* partials, float* out, void* stream);

/* Plan form of tg_guidance_batch (round 3; utils/guidance.py:244-286 per denoising step): the item table refers to the call's
 * tensors by SLOT INDEX; the slot pointers (device memory: attention maps, gradient tensors, box masks, reference columns) are
 * passed per call from HOST memory (`slots_host[n_slots]`, copied into the kernel arguments).  A table depends only on the boxes,
 * token positions, keys, map shapes and loss options, so callers build it once and keep it on the device: a call is then two
 * launches and no host -> device copy (capturable in a hipGraph).  Grid: one block per (item, 4 heads), one head per wave;
 * `head_terms` fp32 [n_items * max_heads] scratch; the fold adds heads in head order and items in item order (bit-identical to
 * tg_guidance_batch and to the per-term entry points).  Same no-shared-(grad, token)-column rule per call. */
#define TG_GUIDANCE_MAX_SLOTS 64
typedef struct {
  int32_t attn_slot, grad_slot /* -1: none */, mask_slot, ref_slot /* -1: none (kind 2 only) */;
  int32_t heads, hw, n_tok, token;
  int32_t kind, k_fg, k_bg, reserved;
  float fg_w, bg_w, scale, eps;
} tg_guidance_pitem;
int tg_guidance_plan_run(const tg_guidance_pitem* items_device, int32_t n_items, int32_t max_hw_topk, int32_t max_heads,
                         const void* const* slots_host, int32_t n_slots, float* head_terms, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Row-chain kernels (round 4, csrc/tg_rowchain.hip): row-local layers at K = C in {320, 640} with the TOKEN ON THE LANE — a wave
 * keeps its 32 token rows in registers as MFMA operands, the weights arrive fragment-packed (theatergen_amd/weights_pack.py::rc_pack)
 * through LDS.  tg_rc_linear: out[m, :] = [LayerNorm-folded] x[m, :] W^T + v (+ res[m, :]), the projections of `Attention`
 * (ip_adapter/attention_processor.py:113-128) and `Transformer2DModel.proj_in / proj_out` (models/transformer_2d.py:150-163) at the
 * first UNet level.  `wpk`: N / 64 chunks of (128 K + 1024) bytes: the chunk's weight fragments, then one vector page (fp32 v[64]:
 * bias, or W beta + bias with the fold; fp32 u[64]: row sums of the rounded W gamma, fold only).  `ln` != 0: x is the UN-normalised
 * stream, statistics per row in fp32 (two-pass, as tg_layernorm).  `variant`: dev switch between workgroup shapes / schedules. */
typedef struct {
  int32_t dtype;
  const void* x; int64_t ldx;
  const void* wpk;
  const void* res; int64_t ldres;
  void* out; int64_t ldc;
  int64_t M;
  int32_t N, K;
  int32_t ln;
  float ln_eps;
  int32_t variant;
  /* reserved (round 4's K = 640 variant, removed in round 5: K must be 320); kept so that the struct layout / ABI version do not move */
  const float* v640;
  const float* u640;
} tg_rc_linear_desc;
int tg_rc_linear(const tg_rc_linear_desc* d, void* stream);

/* tg_rc_xattn: norm2 + attn2 (IPAttnProcessor / AttnProcessor cross-attention) + residual of a first-level BasicTransformerBlock in one
 * launch for SD-1.5's geometry (320 channels = 8 heads x 40, 77 text keys, 0 / 4 / 16 image keys):
 *     out = to_out(softmax(q Kt^T) Vt + w softmax(q Kip^T) Vip) + bias + h,    q = to_q(LayerNorm(h))
 * (models/attention.py:206-224; ip_adapter/attention_processor.py:445-529, 282-393).  `wq` / `wo`: rc_pack chunk streams of the
 * permuted weights (theatergen_amd/rowchain.py: LayerNorm and softmax scale * log2(e) folded into wq); `kv`: tg_rc_kv_pack output
 * [batch][8][24 KiB]; `ip_scale`: DEVICE fp32 scalar (IPAttnProcessor.scale, graph-replayable) or NULL (= 1).
 * tg_rc_kv_pack: text K [batch * text_len, 320] / V^T [batch, 320, ldt] and image K [batch * ip_tokens, 320] / V^T [batch, 320, ldi]
 * (the layouts IPAttnProcessor's projection GEMMs write) -> `out` fragments, batch * 8 * 24 KiB. */
typedef struct {
  int32_t dtype;
  const void* h; int64_t ldh;
  const void* wq;
  const void* kv;
  const void* wo;
  void* out; int64_t ldc;
  int64_t M;
  int32_t rows_per_batch;
  int32_t text_len, ip_tokens;
  float ln_eps;
  const float* ip_scale;
} tg_rc_xattn_desc;
int tg_rc_xattn(const tg_rc_xattn_desc* d, void* stream);
int tg_rc_kv_pack(int32_t dtype, int32_t batch, const void* k, const void* vt, int64_t ldt, int32_t text_len, const void* kip,
                  const void* vtip, int64_t ldi, int32_t ip_tokens, void* out, void* stream);

/* tg_rc_ff: norm3 + FeedForward(GEGLU) + residual of a first-level BasicTransformerBlock (models/attention.py:226-236, 328-338), and with
 * `wpo` != NULL also Transformer2DModel.proj_out + its residual (models/transformer_2d.py:316-327), in one launch:
 *     h3 = net2(a * gelu(g)) + b2 + h,  [a | g] = proj(LayerNorm(h)) + b1;      out = wpo ? proj_out(h3) + b + res0 : h3
 * `w1` / `w2` / `b2`: theatergen_amd/rowchain.py::pack_ff (LayerNorm gamma / beta folded into proj; the kernel normalises the rows);
 * `wpo`: rc_pack_tiles stream of proj_out (+ bias).  `inner` = the hidden width (1280).  `dbg`: dev timing switches, 0. */
typedef struct {
  int32_t dtype;
  const void* h; int64_t ldh;
  const void* w1;
  const void* w2;
  const float* b2;
  const void* wpo;
  const void* res0; int64_t ldres;
  void* out; int64_t ldc;
  int64_t M;
  int32_t inner;
  float ln_eps;
  int32_t dbg;
} tg_rc_ff_desc;
int tg_rc_ff(const tg_rc_ff_desc* d, void* stream);

/* tg_rc_front: the front of a first-level Transformer2DModel in one launch (models/transformer_2d.py:285-296; models/attention.py:186-204):
 *     y = proj_in(GroupNorm(x)) + b;   [Q | K | V] = [to_q ; to_k ; to_v](LayerNorm1(y))
 * `coef`: tg_groupnorm_coef output fp32 [batch][2][320]; `win`: rc_pack_tiles(proj_in, bias); `wqkv`: rc_pack_tiles of the LayerNorm-folded
 * q ; k ; v rows (pack_ln_linear: W', v, u).  Outputs: y [M, 320], Q | K token-major [M, 640] (row pitch ldqk), V transposed per batch item
 * vt[b, c, token] (row pitch ldt) — the operands tg_attention takes. */
typedef struct {
  int32_t dtype;
  const void* x; int64_t ldx;
  const float* coef;
  const void* win;
  const void* wqkv;
  void* y; int64_t ldy;
  void* qk; int64_t ldqk;
  void* vt; int64_t ldt;
  int64_t M;
  int32_t rows_per_batch;
  float ln_eps;
  int32_t dbg;     /* dev timing switches, 0 */
} tg_rc_front_desc;
int tg_rc_front(const tg_rc_front_desc* d, void* stream);

/* tg_skinny_gemm (round 4, csrc/tg_skinny.hip): out[m, :] = act([LayerNorm-folded] x[m, :] W^T + bias) + res[m, :] for a handful of rows — the latent path of
 * the Perceiver Resampler (ip_adapter/resampler.py:13-20 FeedForward, :45-47 to_q / to_kv / to_out, :62-78 forward, :94-97 proj_out: 16 latents x (cond, zero image))
 * and other M <= 32 linears.  `wpk`: theatergen_amd.weights_pack.skinny_pack(W [N, K]) = N / 32 x K / 16 fragment blocks of 64 lanes x 8 elements; the 32 rows are
 * the MFMA B operand in registers, weights go global -> registers in whole-KiB loads, a workgroup = one 32-column tile (8 or 4 waves split K), grid.y = row blocks of 32.
 * Fold (`ln` != 0, K / 64 or K / 128 in {1, 2, 3, 5, 6, 8, 10, 12, 16, 20}): x is the un-normalised stream, W pre-multiplied by gamma, ln_u[n] = row sums of the ROUNDED W gamma,
 * ln_v[n] = W beta (+ bias), statistics two-pass in fp32 (weights_pack.pack_ln_linear).  Epilogue order as tg_gemm: ((acc + bias) + res), act, one rounding.
 * Output routing: `nseg` column segments [seg[i-1].n_end, seg[i].n_end) (multiples of 32), element (m, n) of segment i goes to
 *     ptr + (m / rows_per_batch) * batch_stride + (transposed ? (n - n0) * ld + m % rows_per_batch : (m % rows_per_batch) * ld + (n - n0))
 * — q, the latents' K rows behind the image tokens' rows, and their V^T columns from ONE launch (resampler.py:63-68 `cat((x, latents), dim=-2)` without a copy). */
typedef struct {
  void* ptr;
  int64_t ld;
  int64_t batch_stride;
  int32_t n_end;
  int32_t transposed;
} tg_skinny_seg;
typedef struct {
  int32_t dtype;
  const void* x; int64_t ldx;
  const void* wpk;
  int64_t M;
  int32_t N, K;
  int32_t ln;
  float ln_eps;
  const float* ln_u;
  const float* ln_v;
  const void* bias;        /* storage dtype [N] or NULL */
  int32_t act;             /* TG_ACT_* */
  const void* res; int64_t ldres;
  int32_t nseg;
  int32_t rows_per_batch;
  tg_skinny_seg seg[3];
} tg_skinny_desc;
int tg_skinny_gemm(const tg_skinny_desc* d, void* stream);

/* tg_xq_attn (round 5): norm2 + attn2.to_q + the (decoupled text + image) cross-attention of an inner-level BasicTransformerBlock in ONE launch —
 * reference models/attention.py:206-224 (norm2 -> attn2), ip_adapter/attention_processor.py:282-393 (AttnProcessor) / :396-553 (IPAttnProcessor: two
 * independent softmaxes, :482-516).  The LayerNorm-folded to_q GEMM runs on 128 x 160 tiles (two heads of 80 or one head of 160 channels) and its
 * epilogue computes O = softmax(q Kt^T) Vt + scale * softmax(q Kip^T) Vip from the accumulators: q and the attention launch do not exist; `out` = O [M, C]
 * (what attn.to_out consumes).  `wq` / `ln_u` / `ln_v`: LayerNorm fold of (softmax scale * log2 e) * to_q (host: weights_pack.pack_ln_linear);
 * `kv`: the conditioning's K / V^T as MFMA fragments (tg_xq_kv_pack from the projections tg_gemm writes: k [B * L, C], vt [B, C, ldt], kip, vtip),
 * tg_xq_kv_bytes bytes; `ip_scale`: device scalar or NULL (= 1.0).  head_dim 80 | 160, C % 320 == 0, M and rows_per_batch multiples of 128, text_len <= 96,
 * ip_tokens <= 16. */
typedef struct {
  int32_t dtype;
  const void* x; int64_t ldx;
  const void* wq;
  const float* ln_u;
  const float* ln_v;
  float ln_eps;
  const void* kv;
  const float* ip_scale;
  void* out; int64_t ldc;
  int64_t M;
  int32_t C, head_dim, rows_per_batch, text_len, ip_tokens;
} tg_xq_attn_desc;
int tg_xq_attn(const tg_xq_attn_desc* d, void* stream);
int64_t tg_xq_kv_bytes(int32_t batch, int32_t C, int32_t head_dim);
int tg_xq_kv_pack(int32_t dtype, int32_t batch, int32_t C, int32_t head_dim, const void* k, const void* vt, int64_t ldt, int32_t text_len, const void* kip,
                  const void* vtip, int64_t ldi, int32_t ip_tokens, void* out, void* stream);

/* debugging aid: raw 32x32x16 MFMA on caller-provided fragments (64 lanes x 8 elements each) */
int tg_debug_mfma32(int32_t dtype, const void* a_frags, const void* b_frags, float* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif

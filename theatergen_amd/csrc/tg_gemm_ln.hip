// LayerNorm-fused projections: gemm_glds_kernel<..., LN = true> instances (own translation unit: compile time).
// Replaces the pair  tg_layernorm -> tg_gemm  of BasicTransformerBlock.norm1 -> attn1.to_q|to_k|to_v, norm2 -> attn2.to_q and
// norm3 -> ff.net.0.proj (models/attention.py:186-236): the normalised [M, C] tensor (one HBM write + one read) and the
// layernorm launch disappear; see the kernel comment in tg_gemm_glds.h for the algebra and where the statistics come from.
#include "tg_gemm_glds.h"

namespace {

template <typename T, int STAGES, int BKT, int EPI, int MODE>
int launch_ln_mode(const GemmParams& p, int grid, hipStream_t st) {
  constexpr int BM = 128, BN = 128;
  const size_t lds = (size_t)STAGES * (BM + BN) * BKT * sizeof(T);
  auto k = gemm_glds_kernel<T, BM, BN, 2, 2, false, STAGES, BKT, EPI, MODE>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr;
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, st, p);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

template <typename T, int STAGES, int BKT, int EPI>
int launch_ln(const GemmParams& p, int grid, hipStream_t st) {
  // statistics precomputed by tg_layernorm_stats (ln_rows) or taken from the staged A tiles inside the kernel
  return p.ln_rows != nullptr ? launch_ln_mode<T, STAGES, BKT, EPI, 2>(p, grid, st) : launch_ln_mode<T, STAGES, BKT, EPI, 1>(p, grid, st);
}

// round 5: 128 x 160 tiles (four waves of 32 tokens x 160 channels) where they fill whole rounds of the chip and 128 x 128 does not (tg_gemm.hip: make_plan):
// attn2.to_q of the 32 x 32 / 16 x 16 levels (16384 x 640, 4096 x 1280: 512 / 256 tiles) and attn1's q | k | v at 32 x 32 (16384 x 1920: 1536 = 3 x 512)
template <typename T, int STAGES, int MODE>
int launch_ln160_mode(const GemmParams& p, int grid, hipStream_t st) {
  constexpr int BM = 128, BN = 160, BKT = 64;
  const size_t lds = (size_t)STAGES * (BM + BN) * BKT * sizeof(T);
  auto k = gemm_glds_kernel<T, BM, BN, 4, 1, false, STAGES, BKT, 0, MODE>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr;
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, st, p);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

template <typename T, int STAGES>
int launch_ln160(const GemmParams& p, int grid, hipStream_t st) {
  return p.ln_rows != nullptr ? launch_ln160_mode<T, STAGES, 2>(p, grid, st) : launch_ln160_mode<T, STAGES, 1>(p, grid, st);
}

template <typename T>
int dispatch_ln(const tg_gemm_desc* d, const GemmParams& p, int short_k, int grid, hipStream_t st) {
  if (short_k == 160) return launch_ln160<T, 3>(p, grid, st);        // one workgroup per CU (<= 256 tiles)
  if (short_k == 161) return launch_ln160<T, 2>(p, grid, st);        // two per CU
  if (d->geglu) return short_k ? launch_ln<T, 3, 32, 2>(p, grid, st) : launch_ln<T, 2, 64, 2>(p, grid, st);
  return short_k ? launch_ln<T, 3, 32, 0>(p, grid, st) : launch_ln<T, 2, 64, 0>(p, grid, st);
}

// ---- tg_xq_attn: LayerNorm-folded to_q + cross-attention epilogue (tg_xattn_epi.h) -------------------------------------------------------
template <typename T, int STAGES, int XA>
int launch_xq(const GemmParams& p, int grid, hipStream_t st) {
  constexpr int BM = 128, BN = 160, BKT = 64;
  const size_t lds = (size_t)STAGES * (BM + BN) * BKT * sizeof(T);
  static_assert((size_t)XaGeom<XA>::NPH * 1024 <= (size_t)STAGES * (BM + BN) * BKT * sizeof(T), "a head's K / V^T fragments must fit the dead operand stages");
  auto k = gemm_glds_kernel<T, BM, BN, 4, 1, false, STAGES, BKT, 0, 1, XA>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr;
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, st, p);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

// K / V^T of one conditioning -> the 1-KiB MFMA fragments xattn_epilogue streams.  Block = one fragment: (batch item, 160-column tile, piece).
// Piece order per head: K (key block kb = 0..2 text, 3 image; k-step s) then V^T (row block db of the head; key k-step ks = 0..5 text, 6 image).
template <typename T, int D>
__global__ __launch_bounds__(64) void xq_kv_pack_kernel(const T* k, const T* vt, long ldt, const T* kip, const T* vtip, long ldi, int L, int Tn, int C, T* out) {
  typedef typename Vec<T>::v8 V8;
  typedef XaGeom<D> G;
  const int tiles = C / 160;
  const int piece = blockIdx.x % G::NPT, tn = (blockIdx.x / G::NPT) % tiles, b = blockIdx.x / (G::NPT * tiles);
  const int h = piece / G::NPH, q = piece % G::NPH;
  const int lane = threadIdx.x, i = lane & 31, hi = lane >> 5;
  const int cb = 160 * tn + D * h;                               // first channel of the head
  V8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = from_f32<T>(0.f);
  if (q < 4 * G::KS) {
    const int kb = q / G::KS, s = q % G::KS;
    const bool ip = kb == 3;
    const int key = (ip ? 0 : 32 * kb) + xa_swap23(i);
    if (key < (ip ? Tn : L)) {
      const T* src = (ip ? kip + ((long)b * Tn + key) * C : k + ((long)b * L + key) * C) + cb + 16 * s + 8 * hi;
      o = *reinterpret_cast<const V8*>(src);
    }
  } else {
    const int db = (q - 4 * G::KS) / 7, ks = (q - 4 * G::KS) % 7;
    const bool ip = ks == 6;
    const int ch = (D == 80 ? 32 * (2 * h + db) : 32 * db) + i;   // tile-local channel row of the O^T block
    if (ch >= D * h && ch < D * h + D) {                           // (D = 80: the shared block takes zero rows from the other head)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int key = (ip ? 0 : 16 * ks) + 8 * hi + j;
        if (key < (ip ? Tn : L)) o[j] = ip ? vtip[((long)b * C + 160 * tn + ch) * ldi + key] : vt[((long)b * C + 160 * tn + ch) * ldt + key];
      }
    }
  }
  *reinterpret_cast<V8*>(out + ((long)blockIdx.x * 64 + lane) * 8) = o;
}

}  // namespace

extern "C" int64_t tg_xq_kv_bytes(int32_t batch, int32_t C, int32_t head_dim) {
  if (batch <= 0 || C <= 0 || C % 160 != 0 || (head_dim != 80 && head_dim != 160)) return -1;
  return (int64_t)batch * (C / 160) * (head_dim == 80 ? XaGeom<80>::NPT : XaGeom<160>::NPT) * 1024;
}

extern "C" int tg_xq_kv_pack(int32_t dtype, int32_t batch, int32_t C, int32_t head_dim, const void* k, const void* vt, int64_t ldt, int32_t text_len,
                             const void* kip, const void* vtip, int64_t ldi, int32_t ip_tokens, void* out, void* stream) {
  TG_CHECK(dtype == TG_BF16 || dtype == TG_F16, TG_ERR_ARG, "tg_xq_kv_pack: dtype %d", dtype);
  TG_CHECK(batch > 0 && k && vt && out && C > 0 && C % 160 == 0 && (head_dim == 80 || head_dim == 160), TG_ERR_ARG, "tg_xq_kv_pack: null operand, C = %d, head_dim = %d", C, head_dim);
  TG_CHECK(text_len > 0 && text_len <= 96 && ldt >= text_len && ip_tokens >= 0 && ip_tokens <= 16 && (ip_tokens == 0 || (kip && vtip && ldi >= ip_tokens)), TG_ERR_ARG,
           "tg_xq_kv_pack: %d text keys (<= 96), %d image keys (<= 16)", text_len, ip_tokens);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int npt = head_dim == 80 ? XaGeom<80>::NPT : XaGeom<160>::NPT;
  const unsigned grid = (unsigned)batch * (C / 160) * npt;
#define TG_XQP(TT, DD) hipLaunchKernelGGL((xq_kv_pack_kernel<TT, DD>), dim3(grid), dim3(64), 0, st, (const TT*)k, (const TT*)vt, (long)ldt, (const TT*)kip, \
                                          (const TT*)vtip, (long)ldi, text_len, ip_tokens, C, (TT*)out)
  if (dtype == TG_BF16) { if (head_dim == 80) TG_XQP(bf16_t, 80); else TG_XQP(bf16_t, 160); }
  else { if (head_dim == 80) TG_XQP(f16_t, 80); else TG_XQP(f16_t, 160); }
#undef TG_XQP
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_xq_attn(const tg_xq_attn_desc* d, void* stream) {
  TG_CHECK(d != nullptr, TG_ERR_ARG, "tg_xq_attn: null descriptor");
  TG_CHECK(d->dtype == TG_BF16 || d->dtype == TG_F16, TG_ERR_ARG, "tg_xq_attn: dtype %d", d->dtype);
  TG_CHECK(d->x && d->wq && d->ln_u && d->ln_v && d->kv && d->out && d->ln_eps > 0.f, TG_ERR_ARG, "tg_xq_attn: null operand or eps <= 0");
  TG_CHECK(d->head_dim == 80 || d->head_dim == 160, TG_ERR_UNSUPPORTED, "tg_xq_attn: head_dim %d (80 or 160: two heads / one head per 160-column tile)", d->head_dim);
  TG_CHECK(d->C > 0 && d->C % 160 == 0 && d->C % 64 == 0 && d->C % d->head_dim == 0, TG_ERR_ARG, "tg_xq_attn: C = %d must be a multiple of 320", d->C);
  TG_CHECK(d->M > 0 && d->M % 128 == 0 && d->rows_per_batch > 0 && d->rows_per_batch % 128 == 0 && d->M % d->rows_per_batch == 0, TG_ERR_ARG,
           "tg_xq_attn: M = %lld, rows_per_batch = %d: a 128-token tile must lie inside one batch item", (long long)d->M, d->rows_per_batch);
  TG_CHECK(d->text_len > 0 && d->text_len <= 96 && d->ip_tokens >= 0 && d->ip_tokens <= 16, TG_ERR_ARG, "tg_xq_attn: %d text keys (<= 96), %d image keys (<= 16)",
           d->text_len, d->ip_tokens);
  TG_CHECK(d->ldx == d->C && d->ldc % 8 == 0 && d->ldc >= d->C, TG_ERR_ARG, "tg_xq_attn: x must be [M, C] contiguous, out pitch a multiple of 8");
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  TG_CHECK(al16(d->x) && al16(d->wq) && al16(d->ln_u) && al16(d->ln_v) && al16(d->kv) && al16(d->out), TG_ERR_ARG, "tg_xq_attn: operands must be 16-byte aligned");
  GemmParams p{};
  p.a0 = d->x; p.c0 = d->C; p.w = d->wq; p.M = d->M; p.N = d->C; p.K = d->C; p.lda = d->C; p.ldw = d->C;
  p.rows_per_batch = d->M; p.out = d->out; p.ldc = d->ldc; p.out_scale = 1.0f; p.act = TG_ACT_NONE;
  p.ln_u = d->ln_u; p.ln_v = d->ln_v; p.ln_eps = d->ln_eps;
  const int tiles_n = d->C / 160;
  const long tiles = (d->M / 128) * tiles_n;
  p.full_tiles = (int)tiles; p.tail_s = 1; p.tiles_n = tiles_n; p.tile_bm = 128; p.tile_bn = 160; p.kt_per_split = 0;
  p.epi_lds = 1;
  p.xa_kv = d->kv; p.xa_ip_scale = d->ip_scale; p.xa_rows_per_batch = d->rows_per_batch; p.xa_tiles_n = tiles_n; p.xa_L = d->text_len; p.xa_T = d->ip_tokens;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // tile count decides the instance like the planner's 128 x 160 rule: up to one round of 256 -> three K stages (one workgroup per CU), else two per CU;
  // head_dim 160 needs the three-stage instance (75 KiB of fragments per head)
  const bool three = d->head_dim == 160 || tiles <= 256;
  if (d->dtype == TG_BF16) {
    if (d->head_dim == 80) return three ? launch_xq<bf16_t, 3, 80>(p, (int)tiles, st) : launch_xq<bf16_t, 2, 80>(p, (int)tiles, st);
    return launch_xq<bf16_t, 3, 160>(p, (int)tiles, st);
  }
  if (d->head_dim == 80) return three ? launch_xq<f16_t, 3, 80>(p, (int)tiles, st) : launch_xq<f16_t, 2, 80>(p, (int)tiles, st);
  return launch_xq<f16_t, 3, 160>(p, (int)tiles, st);
}

namespace {
}  // namespace

// short_k: 1 = three 32-wide K stages (three workgroups per CU) instead of two 64-wide ones — the planner's K <= 640 rule; 160 / 161 = 128 x 160 tiles
int tg_gemm_ln_launch(const tg_gemm_desc* d, const void* params, int short_k, int grid, void* stream) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == TG_BF16) return dispatch_ln<bf16_t>(d, p, short_k, grid, st);
  return dispatch_ln<f16_t>(d, p, short_k, grid, st);
}

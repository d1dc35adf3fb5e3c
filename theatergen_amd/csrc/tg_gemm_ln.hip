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

}  // namespace

// short_k: 1 = three 32-wide K stages (three workgroups per CU) instead of two 64-wide ones — the planner's K <= 640 rule; 160 / 161 = 128 x 160 tiles
int tg_gemm_ln_launch(const tg_gemm_desc* d, const void* params, int short_k, int grid, void* stream) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == TG_BF16) return dispatch_ln<bf16_t>(d, p, short_k, grid, st);
  return dispatch_ln<f16_t>(d, p, short_k, grid, st);
}

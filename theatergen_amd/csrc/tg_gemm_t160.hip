// gemm_glds_kernel on 128 x 160 tiles (round 5; own translation unit: compile time).  Plain GEMM, linear epilogue (bias / per-batch vector / residual / scale).
// Four waves of 32 token rows x 160 channels (1 x 5 MFMA tiles per wave, fragments read per k-step).  What the shape is for is the TILE COUNT of the UNet's
// mid-level projections (M x N = 16384 x 640 and 4096 x 1280: models/attention.py:186-236 to_out / net.2, models/transformer_2d.py:285-327 proj_in / proj_out):
// 512 / 256 tiles = exactly two / one per CU, where 128 x 128 gives 640 / 320 tiles on 768 / 512 co-resident slots (tg_gemm.hip: make_plan).
#include "tg_gemm_glds.h"

namespace {

template <typename T, int STAGES, int BKT>
int launch_t160(const GemmParams& p, int grid, hipStream_t st) {
  constexpr int BM = 128, BN = 160;
  const size_t lds = (size_t)STAGES * (BM + BN) * BKT * sizeof(T);
  auto k = gemm_glds_kernel<T, BM, BN, 4, 1, false, STAGES, BKT, 0>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr;
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, st, p);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

template <typename T>
int dispatch_t160(const GemmParams& p, int variant, int grid, hipStream_t st) {
  switch (variant) {
    case 0: return launch_t160<T, 3, 64>(p, grid, st);       // 108 KB: one workgroup per CU, two K-tiles in flight
    default: return launch_t160<T, 2, 64>(p, grid, st);      // 72 KB: two per CU, one K-tile in flight each
  }
}

}  // namespace

int tg_gemm_t160_launch(const tg_gemm_desc* d, const void* params, int variant, int grid, void* stream) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == TG_BF16) return dispatch_t160<bf16_t>(p, variant, grid, st);
  return dispatch_t160<f16_t>(p, variant, grid, st);
}

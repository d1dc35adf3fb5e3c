// Common device helpers for the TheaterGen gfx950 kernels (CDNA4 only: wave64, MFMA, 160 KB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/theatergen_hip.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define TG_WAVE 64

// ---- dtype traits -------------------------------------------------------------------------------
template <typename T> struct Vec;
template <> struct Vec<bf16_t> { typedef bf16x8 v8; typedef bf16x4 v4; };
template <> struct Vec<f16_t> { typedef f16x8 v8; typedef f16x4 v4; };

template <typename T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// 32x32 MFMA C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ int mfma32_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// SiLU = x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 instructions):
// the GroupNorm apply pass spends as long on VALU as on memory, every instruction per element counts.
__device__ __forceinline__ float silu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float quick_gelu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x));
}
// one switch for the generic epilogues / tg_act
__device__ __forceinline__ float gelu_erf_f(float x);
__device__ __forceinline__ float apply_act(float x, int act) {
  if (act == 1) return silu_f(x);
  if (act == 2) return gelu_erf_f(x);
  if (act == 3) return quick_gelu_f(x);
  return x;
}
// exact-erf GELU (F.gelu default, models/attention.py:337).  erfc(|z|) by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7,
// far below bf16 / fp16 resolution): 1 rcp + 5 fma + 1 exp instead of libm's branchy erff — the fused GEGLU epilogue
// runs this on every element of the largest GEMMs.  The negative branch uses erfc directly (no 1 - (1 - tiny)).
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float z = __builtin_fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  poly = __builtin_fmaf(poly, t, 1.421413741f);
  poly = __builtin_fmaf(poly, t, -0.284496736f);
  poly = __builtin_fmaf(poly, t, 0.254829592f);
  const float half_erfc = 0.5f * poly * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);   // 0.5 * erfc(|z|)
  return x * (x >= 0.f ? 1.0f - half_erfc : half_erfc);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- host-side error plumbing (tg_api.hip) -------------------------------------------------------
void tg_set_error(const char* fmt, ...);
#define TG_CHECK(cond, code, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      tg_set_error(__VA_ARGS__);             \
      return (code);                         \
    }                                        \
  } while (0)
#define TG_LAUNCH_CHECK()                                                \
  do {                                                                   \
    hipError_t e__ = hipGetLastError();                                  \
    if (e__ != hipSuccess) {                                             \
      tg_set_error("HIP launch failed: %s", hipGetErrorString(e__));     \
      return TG_ERR_LAUNCH;                                              \
    }                                                                    \
  } while (0)

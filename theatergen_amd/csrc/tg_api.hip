// Error plumbing + version for libtheatergen_hip.so (C ABI: include/theatergen_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "tg_common.h"

namespace {
thread_local char g_err[512] = "";
}

void tg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* tg_last_error(void) { return g_err; }
extern "C" int tg_version(void) { return TG_ABI_VERSION; }

template <typename T>
__global__ void debug_mfma32_kernel(const typename Vec<T>::v8* a, const typename Vec<T>::v8* b, f32x16* d) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = mfma32(a[threadIdx.x], b[threadIdx.x], acc);
  d[threadIdx.x] = acc;
}

extern "C" int tg_debug_mfma32(int32_t dtype, const void* a_frags, const void* b_frags, float* d_out, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == TG_BF16)
    hipLaunchKernelGGL(debug_mfma32_kernel<bf16_t>, dim3(1), dim3(64), 0, st, (const bf16x8*)a_frags, (const bf16x8*)b_frags, (f32x16*)d_out);
  else
    hipLaunchKernelGGL(debug_mfma32_kernel<f16_t>, dim3(1), dim3(64), 0, st, (const f16x8*)a_frags, (const f16x8*)b_frags, (f32x16*)d_out);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

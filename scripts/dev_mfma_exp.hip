// dev probe (not part of the product): do the matrix pipe and the VALU's transcendental unit run CONCURRENTLY on a gfx950 SIMD?  The self-attention kernel's
// tile costs the SUM of its MFMA and VALU time; this isolates the question from that kernel's dependencies:
//   per iteration NM independent v_mfma_f32_32x32x16_bf16 (4 accumulator chains) and NE independent v_exp_f32 (inputs fixed, outputs discarded),
//   layout 0: all MFMAs, then all exps;  layout 1: NE / NM exps behind every MFMA (same wave, program order interleaved);
//   layout 2: role split — even waves of a SIMD only MFMAs, odd waves only exps (cross-wave co-issue);  KIND 1 = v_fma_f32 (full rate) instead of v_exp_f32.
// One workgroup per CU (grid 256), THREADS / 256 waves per SIMD.  Host prints ns per iteration: T(NM, 0), T(0, NE), T(NM, NE) -> sum or max.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/dev_mfma_exp.hip -o scripts/_build/libmfma_exp.so
#include <hip/hip_runtime.h>
#include <cstdint>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NM, int NE, int LAYOUT, int KIND, int THREADS>
__global__ __launch_bounds__(THREADS) void probe_kernel(const float* __restrict__ src, int rounds, float* out) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  bf16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)src[(tid * 8 + j) & 4095]; b[j] = (__bf16)src[(tid * 8 + j + 77) & 4095]; }
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x[16], y[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { x[i] = src[(tid + 64 * i) & 4095]; y[i] = 0.f; }
  const bool do_m = LAYOUT != 2 || (wave >> 2 & 1) == 0;       // waves 0-3 -> SIMDs 0-3, waves 4-7 -> SIMDs 0-3 again: bit 2 = which of the SIMD's waves
  const bool do_e = LAYOUT != 2 || (wave >> 2 & 1) == 1;
  auto vop = [&](int i) {
    if constexpr (KIND == 0) asm volatile("v_exp_f32 %0, %1" : "=v"(y[i & 15]) : "v"(x[i & 15]));
    else asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(y[i & 15]) : "v"(x[i & 15]));
  };
  for (int it = 0; it < rounds; ++it) {
    if constexpr (LAYOUT == 1) {
      constexpr int EPM = NM > 0 ? NE / NM : 0;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < EPM; ++e) vop(m * EPM + e);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      if (do_m) {
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (do_e) {
#pragma unroll
        for (int e = 0; e < NE; ++e) vop(e);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += y[i];
  if (s == 12345.678f) out[0] = s;
}

template <int NM, int NE, int LAYOUT, int KIND, int THREADS>
static int run(const float* src, int rounds, float* out, hipStream_t st) {
  hipLaunchKernelGGL((probe_kernel<NM, NE, LAYOUT, KIND, THREADS>), dim3(256), dim3(THREADS), 0, st, src, rounds, out);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

#define CASES(NM, NE)                                                                       \
  if (nm == NM && ne == NE) {                                                               \
    if (threads == 256) {                                                                   \
      if (layout == 0 && kind == 0) return run<NM, NE, 0, 0, 256>(src, rounds, out, st);    \
      if (layout == 1 && kind == 0) return run<NM, NE, 1, 0, 256>(src, rounds, out, st);    \
      if (layout == 0 && kind == 1) return run<NM, NE, 0, 1, 256>(src, rounds, out, st);    \
      if (layout == 1 && kind == 1) return run<NM, NE, 1, 1, 256>(src, rounds, out, st);    \
    } else if (threads == 512) {                                                            \
      if (layout == 0 && kind == 0) return run<NM, NE, 0, 0, 512>(src, rounds, out, st);    \
      if (layout == 1 && kind == 0) return run<NM, NE, 1, 0, 512>(src, rounds, out, st);    \
      if (layout == 2 && kind == 0) return run<NM, NE, 2, 0, 512>(src, rounds, out, st);    \
      if (layout == 2 && kind == 1) return run<NM, NE, 2, 1, 512>(src, rounds, out, st);    \
    } else if (threads == 768) {                                                            \
      if (layout == 0 && kind == 0) return run<NM, NE, 0, 0, 768>(src, rounds, out, st);    \
      if (layout == 1 && kind == 0) return run<NM, NE, 1, 0, 768>(src, rounds, out, st);    \
    }                                                                                       \
  }

extern "C" int mfma_exp_probe(int nm, int ne, int layout, int kind, int threads, const float* src, int rounds, float* out, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  CASES(8, 0) CASES(0, 16) CASES(8, 16) CASES(8, 8) CASES(0, 8) CASES(8, 32) CASES(0, 32)
  return 2;
}

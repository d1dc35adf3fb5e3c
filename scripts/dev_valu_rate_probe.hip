// dev probe (not part of the product): issue cost of the softmax loop's VALU instructions on gfx950, per wave64
// instruction, with W waves per SIMD, alone and next to waves that only issue MFMAs — is v_exp_f32 quarter rate, and do
// transcendental / VALU work of one wave and MFMA work of another overlap on a SIMD?
// Build: hipcc --offload-arch=gfx950 -O3 scripts/dev_valu_rate_probe.hip -o scripts/_build/valuprobe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// KIND: 0 v_exp_f32, 1 v_fma_f32, 2 v_pk_fma_f32, 3 v_cvt_pk_bf16_f32, 4 v_max3_f32, 5 MFMA 32x32x16 bf16,
//       6 mix per 'tile': 32 fma + 32 exp + 16 cvt + 16 max3 (the softmax loop), 7 = 6 in waves 0..1, MFMA in waves 2..3
template <int KIND>
__global__ __launch_bounds__(1024) void rate(int iters, float seed, float* out, long long* cyc) {
  float x[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] = seed * (float)(i + 1) - 1.0f - (float)threadIdx.x * 1e-3f;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  bf16x8 af, bfr;
#pragma unroll
  for (int j = 0; j < 8; ++j) { af[j] = (__bf16)seed; bfr[j] = (__bf16)(seed + 1.f); }
  const int wave = threadIdx.x >> 6;
  const bool mfma_wave = KIND == 5 || (KIND == 7 && (wave & 3) >= 2);     // waves are dealt round-robin to the 4 SIMDs:
  // waves w, w+4, w+8 ... share a SIMD; (wave >> 2) is the slot on the SIMD -> use that for the split instead
  const bool mf = KIND == 5 || (KIND == 7 && ((wave >> 2) & 1));
  (void)mfma_wave;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (mf) {
#pragma unroll
      for (int i = 0; i < 14; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc, 0, 0, 0);
    } else if (KIND == 0) {
#pragma unroll
      for (int i = 0; i < 32; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]);
    } else if (KIND == 1) {
#pragma unroll
      for (int i = 0; i < 32; ++i) x[i] = __builtin_fmaf(x[i], seed, 0.25f);
    } else if (KIND == 2) {
#pragma unroll
      for (int i = 0; i < 32; i += 2) {
        f32x2 v = {x[i], x[i + 1]};
        f32x2 s = {seed, seed}, c = {0.25f, 0.5f};
        v = __builtin_elementwise_fma(v, s, c);
        x[i] = v[0]; x[i + 1] = v[1];
      }
    } else if (KIND == 3) {
#pragma unroll
      for (int i = 0; i < 32; i += 2) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
        bf2 r = {(__bf16)x[i], (__bf16)x[i + 1]};
        x[i] = __builtin_bit_cast(float, r) ; x[i + 1] += 1.f;
      }
    } else if (KIND == 4) {
#pragma unroll
      for (int i = 0; i < 32; i += 2) x[i] = fmaxf(fmaxf(x[i], x[i + 1]), x[(i + 2) & 31]);
    } else {
      float mx = -1e30f;
#pragma unroll
      for (int i = 0; i < 32; i += 2) mx = fmaxf(fmaxf(mx, x[i]), x[i + 1]);
#pragma unroll
      for (int i = 0; i < 32; ++i) x[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], seed, -mx * seed));
#pragma unroll
      for (int i = 0; i < 32; i += 2) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
        bf2 r = {(__bf16)x[i], (__bf16)x[i + 1]};
        x[i] += __builtin_bit_cast(float, r);
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += x[i];
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc[r];
  if (s == 123.456f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND>
void run(const char* name, int per_iter, float* out, long long* cyc) {
  for (int waves_per_simd : {1, 2, 3, 4}) {
    const int threads = 64 * 4 * waves_per_simd;
    const int iters = 2000;
    hipLaunchKernelGGL(rate<KIND>, dim3(256), dim3(threads), 0, 0, iters, 0.37f, out, cyc);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate<KIND>, dim3(256), dim3(threads), 0, 0, iters, 0.37f, out, cyc);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long c = 0;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // s_memtime ticks at 100 MHz: derive core cycles from the event time at an assumed 2.4 GHz too
    const double cyc_evt = ms * 1e-3 * 2.4e9 / iters;
    printf("%-34s waves/SIMD %d: %8.1f us  ~%7.1f clk/iter @2.4GHz  (%5.2f clk per instr per wave)\n", name, waves_per_simd,
           ms * 1e3, cyc_evt, cyc_evt / per_iter / waves_per_simd);
  }
}

int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 8);
  run<0>("v_exp_f32 x32", 32, out, cyc);
  run<1>("v_fma_f32 x32", 32, out, cyc);
  run<2>("v_pk_fma_f32 x16", 16, out, cyc);
  run<3>("cvt_pk_bf16 x16 (+16 add)", 32, out, cyc);
  run<4>("v_max3 x16", 16, out, cyc);
  run<5>("mfma 32x32x16 bf16 x14", 14, out, cyc);
  run<6>("softmax mix (32 fma 32 exp 16 cvt 16 max)", 1, out, cyc);
  run<7>("mix waves + mfma waves (half/half)", 1, out, cyc);
  return 0;
}

// dev probe (not part of the product): what the matrix pipe SUSTAINS on this part on random bf16 operands, every CU busy, for seconds —
// the yardstick `roofline.sustained_peak` of bench.py is read from (profiles/r6_mfma_sustained.json, written by scripts/r6_mfma_sustained.py).
// Three bodies, all 256 CUs, 2 waves per SIMD (512-thread workgroups, one per CU) or 1 wave per SIMD (256-thread):
//   mode 0: MFMA only — NA x NB register-resident fragments (loaded once from a random buffer), NA * NB independent 32x32 accumulators, back to back
//   mode 1: + the fragments are re-read from LDS (random bytes) every round: the ds_read_b128 traffic of a GEMM K loop (NA + NB reads per NA * NB MFMAs)
//   mode 2: mode 1 + LDS-DMA refills of the LDS image from a (L2-resident) global buffer at a GEMM's bytes-per-MFMA rate
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/dev_mfma_sustained.hip -o scripts/_build/libmfma_sustained.so
#include <hip/hip_runtime.h>
#include <cstdint>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int NA, int NB, int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void sustain_kernel(const u32x4* __restrict__ src, long src_vecs, int rounds, float* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = THREADS / 64;
  // LDS image: per wave (NA + NB) fragments x 2 phases of 1 KiB (64 lanes x 16 B), filled from the random buffer
  constexpr int PH = 2;
  constexpr int PER_WAVE = (NA + NB) * PH * 1024;
  char* my = lds + wave * PER_WAVE;
  const long base = ((long)blockIdx.x * NW + wave) * (NA + NB) * PH * 64 + lane;
  for (int f = 0; f < (NA + NB) * PH; ++f) {
    const u32x4 v = src[(base + (long)f * 64) % src_vecs];
    *reinterpret_cast<u32x4*>(my + f * 1024 + lane * 16) = v;
  }
  __syncthreads();
  bf16x8 a[NA], b[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) a[i] = *reinterpret_cast<const bf16x8*>(my + i * 1024 + lane * 16);
#pragma unroll
  for (int j = 0; j < NB; ++j) b[j] = *reinterpret_cast<const bf16x8*>(my + (NA + j) * 1024 + lane * 16);
  f32x16 acc[NA][NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const u32x4* gsrc = src + ((long)blockIdx.x * 4096 + tid) % (src_vecs - 65536);
  for (int it = 0; it < rounds; ++it) {
#pragma unroll
    for (int ph = 0; ph < PH; ++ph) {
      if constexpr (MODE >= 1) {
        // the next phase's fragments are requested before this phase's MFMAs (software pipeline, as a K loop would)
        bf16x8 na[NA], nb[NB];
        const char* p = my + ((ph + 1) % PH) * (NA + NB) * 1024 + lane * 16;
#pragma unroll
        for (int i = 0; i < NA; ++i) na[i] = *reinterpret_cast<const bf16x8*>(p + i * 1024);
#pragma unroll
        for (int j = 0; j < NB; ++j) nb[j] = *reinterpret_cast<const bf16x8*>(p + (NA + j) * 1024);
        if constexpr (MODE >= 2) {
          // one 1-KiB LDS-DMA piece per wave and phase into a scratch area behind the fragment image (never read: traffic only)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (long)((it * PH + ph) & 63) * 1024),
                                           (__attribute__((address_space(3))) void*)(lds + NW * PER_WAVE + wave * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NA; ++i) a[i] = na[i];
#pragma unroll
        for (int j = 0; j < NB; ++j) b[j] = nb[j];
      } else {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) out[0] = s;            // keeps the accumulators live; never true on random data in practice
}

template <int NA, int NB, int MODE, int THREADS>
static int launch(const void* src, long src_vecs, int rounds, int blocks, float* out, hipStream_t st) {
  constexpr int NW = THREADS / 64;
  const size_t ldsb = (size_t)NW * (NA + NB) * 2 * 1024 + NW * 1024;
  auto k = sustain_kernel<NA, NB, MODE, THREADS>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  (void)attr;
  hipLaunchKernelGGL(k, dim3(blocks), dim3(THREADS), ldsb, st, reinterpret_cast<const u32x4*>(src), src_vecs, rounds, out);
  return (int)hipGetLastError();
}

// flop per launch = blocks * waves * rounds * 2 phases * NA * NB * 2 * 32 * 32 * 16
extern "C" int probe_sustain(int mode, int waves_per_simd, const void* src, long src_vecs, int rounds, int blocks, float* out, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (waves_per_simd == 2) {
    if (mode == 0) return launch<2, 3, 0, 512>(src, src_vecs, rounds, blocks, out, st);
    if (mode == 1) return launch<2, 3, 1, 512>(src, src_vecs, rounds, blocks, out, st);
    return launch<2, 3, 2, 512>(src, src_vecs, rounds, blocks, out, st);
  }
  if (mode == 0) return launch<2, 4, 0, 256>(src, src_vecs, rounds, blocks, out, st);
  if (mode == 1) return launch<2, 4, 1, 256>(src, src_vecs, rounds, blocks, out, st);
  return launch<2, 4, 2, 256>(src, src_vecs, rounds, blocks, out, st);
}
extern "C" long probe_sustain_flop(int waves_per_simd, int rounds, int blocks) {
  const long per_round = waves_per_simd == 2 ? 8L * 2 * 6 : 4L * 2 * 8;
  return (long)blocks * rounds * per_round * 2 * 32 * 32 * 16;
}

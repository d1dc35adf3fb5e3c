// dev probe (not part of the product): an 8-wave "ping-pong" GEMM main loop for gfx950 — out[M, N] = X[M, K] · W[N, K]^T, bf16, fp32 accumulate.
// Two groups of four waves (one wave of each group per SIMD) run the SAME phase program one barrier apart: while one group issues its MFMA block the
// other reads fragments and issues LDS-DMA, so the matrix pipe of every SIMD alternates between its two waves (cdna_hip_programming.md §5 "8-phase").
//   tile 256 x BN x 64, waves 2 (M) x 4 (N), wave tile 128 x BN/4 of 16x16x32 MFMAs, four phases per K-tile (C quadrants), operands by LDS-DMA into
//   half-tile slots (A rows 0-127 / 128-255, W columns 0-BN/2 / BN/2-BN) x 2 K-tile parities, refilled two K-tiles ahead, counted vmcnt.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/dev_gemm8.hip -o scripts/_build/libgemm8.so
#include <hip/hip_runtime.h>
#include <cstdint>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

__device__ __forceinline__ int xcd_chunk(int bid, int n) {
  const int q = n >> 3, r = n & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// BN: 256 (wave tile 128 x 64: quadrants 64 x 32) or 128 (wave tile 128 x 32: quadrants 64 x 16)
template <int BN, int FLAGS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm8_kernel(const __bf16* __restrict__ X, const __bf16* __restrict__ W,
                                                                                                 __bf16* __restrict__ out, int M, int N, int K, int tiles_n) {
  constexpr int BM = 256, BK = 64;
  constexpr int WNC = BN / 4;                     // columns per wave
  constexpr int QN = WNC / 2;                     // columns per quadrant
  constexpr int TNQ = QN / 16;                    // 16-column MFMA tiles per quadrant (2 or 1)
  constexpr int A_HALF = 128 * 128;               // bytes of one A half-tile (128 rows x 64 k)
  constexpr int B_HALF = (BN / 2) * 128;
  constexpr int PARITY = 2 * A_HALF + 2 * B_HALF; // one K-tile
  constexpr int NA = A_HALF / 8192, NB = B_HALF / 8192;   // LDS-DMA instructions per wave per half-tile (8 waves x 1 KiB each)
  static_assert(NA * 8192 == A_HALF && NB * 8192 == B_HALF, "half-tiles are whole rounds of eight 1-KiB pieces");
  extern __shared__ __attribute__((aligned(128))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int ntiles = (M / BM) * tiles_n;
  const int lb = xcd_chunk(blockIdx.x, ntiles);
  const int tile_m = lb / tiles_n, tile_n = lb - tile_m * tiles_n;
  const long m0 = (long)tile_m * BM, n0 = (long)tile_n * BN;
  const int nkt = K / BK;

  // ---- LDS-DMA lane geometry: piece q (1 KiB) = rows [8q, 8q + 8) of a half-tile; lane -> (row 8q + lane / 8, slot lane % 8); the 16-byte chunk
  // fetched into a slot is slot ^ key(row), key(row) = (row >> 1) & 7
  const int lrow = lane >> 3;
  auto dma = [&](const __bf16* src, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_byte_addr) : "memory");
  };
  // this lane's source pointers (K-tile 0) for its pieces of each half-tile
  const __bf16* asrc[2][NA];
  const __bf16* bsrc[2][NB];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int r = (j * 8 + wave) * 8 + lrow;                   // row inside the half
      const int ch = (lane & 7) ^ ((r >> 1) & 7);
      asrc[h][j] = X + (m0 + h * 128 + r) * (long)K + ch * 8;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int r = (j * 8 + wave) * 8 + lrow;
      const int ch = (lane & 7) ^ ((r >> 1) & 7);
      bsrc[h][j] = W + (n0 + h * (BN / 2) + r) * (long)K + ch * 8;
    }
  }
  // half-tile ids: 0 = A rows 0-127, 1 = A rows 128-255, 2 = W columns 0-BN/2, 3 = the other W half
  auto issue_half = [&](int id, int kt) {
    const unsigned par = (unsigned)(kt & 1) * PARITY;
    const long koff = (long)kt * BK;
    if (id < 2) {
#pragma unroll
      for (int j = 0; j < NA; ++j) dma(asrc[id][j] + koff, lds0 + par + (unsigned)id * A_HALF + (unsigned)(j * 8 + wave) * 1024u);
    } else {
#pragma unroll
      for (int j = 0; j < NB; ++j) dma(bsrc[id - 2][j] + koff, lds0 + par + 2u * A_HALF + (unsigned)(id - 2) * B_HALF + (unsigned)(j * 8 + wave) * 1024u);
    }
  };

  // ---- fragment addresses: lane -> row (lane & 15) of a 16-row tile, chunk 4 ks + (lane >> 4)
  const int frow = lane & 15, fq = lane >> 4;
  // A: half wr, rows qa * 64 + i * 16 + frow  (i = 0..3);  W: half (wc >> 1), rows (wc & 1) * WNC + qb * QN + j * 16 + frow
  auto a_addr = [&](int par, int qa, int i, int ks) -> unsigned {
    const int r = qa * 64 + i * 16 + frow;
    return lds0 + (unsigned)par * PARITY + (unsigned)wr * A_HALF + (unsigned)r * 128u + (unsigned)(((4 * ks + fq) ^ ((r >> 1) & 7)) << 4);
  };
  auto b_addr = [&](int par, int qb, int j, int ks) -> unsigned {
    const int r = (wc & 1) * WNC + qb * QN + j * 16 + frow;
    return lds0 + (unsigned)par * PARITY + 2u * A_HALF + (unsigned)(wc >> 1) * B_HALF + (unsigned)r * 128u + (unsigned)(((4 * ks + fq) ^ ((r >> 1) & 7)) << 4);
  };
  u32x4 af[4][2], bf[TNQ][2];
  auto read_a = [&](int par, int qa) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) asm volatile("ds_read_b128 %0, %1" : "=v"(af[i][ks]) : "v"(a_addr(par, qa, i, ks)));
  };
  auto read_b = [&](int par, int qb) {
#pragma unroll
    for (int j = 0; j < TNQ; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) asm volatile("ds_read_b128 %0, %1" : "=v"(bf[j][ks]) : "v"(b_addr(par, qb, j, ks)));
  };
  f32x4 acc[2][2][4][TNQ];                       // [qa][qb][i][j]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TNQ; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mfma_quad = [&](int qa, int qb) {
    if (FLAGS & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TNQ; ++j)
          acc[qa][qb][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[j][ks]), __builtin_bit_cast(bf16x8, af[i][ks]),
                                                                      acc[qa][qb][i][j], 0, 0, 0);
    if (FLAGS & 1) __builtin_amdgcn_s_setprio(0);
  };
#define LOAD_END()                                         \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();                          \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)
#define MFMA_END()                                         \
  do {                                                     \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();                          \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)

  // ---- prologue: K-tiles 0 and 1 requested; K-tile 0 landed
  issue_half(0, 0); issue_half(1, 0); issue_half(2, 0); issue_half(3, 0);
  if (nkt > 1) { issue_half(0, 1); issue_half(1, 1); issue_half(2, 1); issue_half(3, 1); }
  if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NA + 2 * NB) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();     // the second group runs one barrier behind the first
  // In-loop refill schedule for K-tile t (t >= 0), phase p: p1 -> A half 1 of tile t+1 was the last prologue piece ... steady state:
  //   p4 of tile t:  A half 0 of tile t + 2   (both A halves of parity t were read for the last time in p3)
  //   p1 of tile t+1: A half 1 of tile t + 2
  //   p2 of tile t+1: W half 0 of tile t + 2  (the W halves of parity t were read for the last time in p4 of tile t)
  //   p3 of tile t+1: W half 1 of tile t + 2
  // and the wait at the end of p4's load section (before its first barrier) leaves only p4's own request in flight.
  for (int t = 0; t < nkt; ++t) {
    const int par = t & 1;
    // ---- phase 1: quadrant (0, 0)
    read_b(par, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(par, 0);
    if (t >= 1 && t + 1 < nkt) issue_half(1, t + 1);
    LOAD_END();
    mfma_quad(0, 0);
    MFMA_END();
    // ---- phase 2: quadrant (0, 1)
    read_b(par, 1);
    if (t >= 1 && t + 1 < nkt) issue_half(2, t + 1);
    LOAD_END();
    mfma_quad(0, 1);
    MFMA_END();
    // ---- phase 3: quadrant (1, 1)
    read_a(par, 1);
    if (t >= 1 && t + 1 < nkt) issue_half(3, t + 1);
    LOAD_END();
    mfma_quad(1, 1);
    MFMA_END();
    // ---- phase 4: quadrant (1, 0)
    read_b(par, 0);
    if (t + 2 < nkt) {
      issue_half(0, t + 2);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LOAD_END();
    mfma_quad(1, 0);
    MFMA_END();
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();     // pairs with the second group's last barrier

  // ---- epilogue (probe: direct 8-byte stores; lane = token, 4 consecutive channels per accumulator)
#pragma unroll
  for (int qa = 0; qa < 2; ++qa)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TNQ; ++j) {
          const long m = m0 + wr * 128 + qa * 64 + i * 16 + frow;
          const long n = n0 + wc * WNC + qb * QN + j * 16 + 4 * fq;
          const f32x4 v = acc[qa][qb][i][j];
          typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
          bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
          *reinterpret_cast<bf16x4*>(out + m * N + n) = o;
        }
}

template <int BN, int FLAGS>
static int launch(const void* X, const void* W, void* out, int M, int N, int K, hipStream_t st) {
  constexpr size_t lds = 2 * (2 * 128 * 128 + 2 * (BN / 2) * 128);
  if (M % 256 || N % BN || K % 64) return -1;
  auto k = gemm8_kernel<BN, FLAGS>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr;
  const int tiles_n = N / BN;
  hipLaunchKernelGGL(k, dim3((M / 256) * tiles_n), dim3(512), lds, st, (const __bf16*)X, (const __bf16*)W, (__bf16*)out, M, N, K, tiles_n);
  return (int)hipGetLastError();
}

extern "C" int gemm8(int bn, int flags, const void* X, const void* W, void* out, int M, int N, int K, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (bn == 256) return flags & 1 ? launch<256, 1>(X, W, out, M, N, K, st) : launch<256, 0>(X, W, out, M, N, K, st);
  if (bn == 128) return flags & 1 ? launch<128, 1>(X, W, out, M, N, K, st) : launch<128, 0>(X, W, out, M, N, K, st);
  return -2;
}

// dev probe (not part of the product): speed of light of the GEMM inner-loop STRUCTURE on gfx950, built up in layers:
// MFMA only -> + LDS fragment reads -> + one barrier per K-tile -> + LDS-DMA operand tiles from a warm L2.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/dev_mfma_loop_probe.hip -o scripts/_build/mfmaprobe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// flags: 16 = skip the MFMAs, 32 = 3 LDS stages / 2 tiles in flight, 64 = DMA instructions spread between the k-steps
// flags: 1 = ds_read fragments every K-tile, 2 = barrier per K-tile, 4 = LDS-DMA the next tile, 8 = k-step software pipeline
template <int TM, int TN, int FLAGS>
__global__ __launch_bounds__(256) void loopk(const char* __restrict__ src, long ld, int ktiles, int iters, float* out) {
  constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32, ROWS = BM + BN, NDMA = ROWS / 32;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wave_m = wave >> 1, wave_n = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5, rkey = (l31 >> 1) & 7;
  const long xrow = (long)(b % 8) * BM, wrow = (long)(8 * BM) + (long)((b / 8) % 8) * BN;
  constexpr int ST = (FLAGS & 32) ? 3 : 2;
  auto issue_part = [&](int t, int slot, int j0, int j1) {
    const int kt = t % ktiles;
#pragma unroll
    for (int j = j0; j < j1; ++j) {
      const int piece = j * 256 + tid;
      const int r = piece >> 3, c = piece & 7;
      const long row = r < BM ? xrow + r : wrow + (r - BM);
      const char* g = src + row * ld + (long)kt * 128 + ((c ^ ((r >> 1) & 7)) * 16);
      char* l = lds + (size_t)slot * ROWS * 128 + (size_t)(j * 256 + (tid & ~63)) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    }
  };
  auto issue = [&](int t, int slot) { issue_part(t, slot, 0, NDMA); };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int i = tid; i < ST * ROWS * 128 / 4; i += 256) reinterpret_cast<int*>(lds)[i] = 0x3c003c00;
  __syncthreads();
  if (FLAGS & 4) {
    issue(0, 0);
    if (ST == 3) issue(1, 1);
    if (ST == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  bf16x8 xf[4][TM], wf[4][TN];
  auto read_ks = [&](int buf, int ks) {
    const char* bx = lds + (size_t)buf * ROWS * 128 + (size_t)(wave_m * TM * 32 + l31) * 128;
    const char* bw = lds + (size_t)buf * ROWS * 128 + (size_t)(BM + wave_n * TN * 32 + l31) * 128;
    const int so = ((2 * ks + hi) ^ rkey) * 16;
#pragma unroll
    for (int i = 0; i < TM; ++i) xf[ks][i] = *reinterpret_cast<const bf16x8*>(bx + i * 32 * 128 + so);
#pragma unroll
    for (int j = 0; j < TN; ++j) wf[ks][j] = *reinterpret_cast<const bf16x8*>(bw + j * 32 * 128 + so);
  };
  auto mma_ks = [&](int ks) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (FLAGS & 16) acc[i][j][0] += (float)wf[ks][j][0] * (float)xf[ks][i][0];
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][j], xf[ks][i], acc[i][j], 0, 0, 0);
      }
  };
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) read_ks(0, ks);
  int buf = 0;
  for (int it = 0; it < iters; ++it) {
    if (FLAGS & 8) {
      // k-step pipeline: fragments of step ks+1 are requested before the MFMAs of step ks are issued
      if (FLAGS & 1) read_ks(buf, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (FLAGS & 4) issue(it + ST - 1, (buf + ST - 1) % ST);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if ((FLAGS & 1) && ks + 1 < 4) read_ks(buf, ks + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_ks(ks);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      if (FLAGS & 1) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) read_ks(buf, ks);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int nb = (buf + ST - 1) % ST;
      if ((FLAGS & 4) && !(FLAGS & 64)) issue(it + ST - 1, nb);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if ((FLAGS & 4) && (FLAGS & 64)) issue_part(it + ST - 1, nb, ks * NDMA / 4, (ks + 1) * NDMA / 4);
        __builtin_amdgcn_sched_barrier(0);
        mma_ks(ks);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (FLAGS & 4) {
      if (ST == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (FLAGS & 2) __builtin_amdgcn_s_barrier();
    if (FLAGS & 4) buf = (buf + 1) % ST;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.f) out[0] = s;
}

template <int TM, int TN, int FLAGS>
void run(const char* src, long ld, int ktiles, int bpc, float* out, const char* what) {
  constexpr int ROWS = 2 * TM * 32 + 2 * TN * 32;
  const int iters = 1000;
  const size_t need = ((FLAGS & 32) ? 3 : 2) * ROWS * 128;
  size_t alloc = 160 * 1024 / bpc - 512;
  if (alloc > 160 * 1024 - 1024) alloc = 160 * 1024 - 1024;
  if (need > alloc) return;
  (void)hipFuncSetAttribute((const void*)loopk<TM, TN, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)alloc);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int grid = 256 * bpc;
  loopk<TM, TN, FLAGS><<<grid, 256, alloc>>>(src, ld, ktiles, iters, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  loopk<TM, TN, FLAGS><<<grid, 256, alloc>>>(src, ld, ktiles, iters, out);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double fl = 2.0 * (2 * TM * 32) * (2 * TN * 32) * 64 * (double)iters * grid;
  printf("wave %3dx%-3d blocks/CU=%d %-34s %7.0f TF  (%.0f cyc/K-tile @2.4GHz)  %s\n", TM * 32, TN * 32, bpc, what, fl / ms / 1e9,
         ms * 1e-3 / iters * 2.4e9, hipGetErrorString(hipGetLastError()));
}

template <int TM, int TN>
void suite(const char* src, long ld, int kt, float* out) {
  for (int bpc : {1, 2}) {
    run<TM, TN, 0>(src, ld, kt, bpc, out, "mfma only");
    run<TM, TN, 1 | 2>(src, ld, kt, bpc, out, "reads+barrier");
    run<TM, TN, 1 | 2 | 4>(src, ld, kt, bpc, out, "reads+barrier+dma");
    run<TM, TN, 1 | 2 | 4 | 16>(src, ld, kt, bpc, out, "reads+barrier+dma, NO mfma");
    run<TM, TN, 2 | 4 | 16>(src, ld, kt, bpc, out, "barrier+dma only");
    run<TM, TN, 2 | 4>(src, ld, kt, bpc, out, "mfma+barrier+dma, no reads");
    run<TM, TN, 2 | 4 | 64>(src, ld, kt, bpc, out, "mfma+barrier+dma(spread), no reads");
    run<TM, TN, 1 | 2 | 4 | 64>(src, ld, kt, bpc, out, "reads+barrier+dma(spread)");
    run<TM, TN, 2 | 4 | 32>(src, ld, kt, bpc, out, "mfma+barrier+dma 3-stage, no reads");
    run<TM, TN, 1 | 2 | 4 | 32>(src, ld, kt, bpc, out, "reads+barrier+dma 3-stage");
    run<TM, TN, 1 | 2 | 4 | 32 | 64>(src, ld, kt, bpc, out, "reads+barrier+dma(spread) 3-stage");
  }
}

int main() {
  const long ld = 2560;
  const int rows = 16 * 256 * 2, ktiles = 20;
  char* src;
  float* out;
  (void)hipMalloc(&src, (size_t)rows * ld);
  (void)hipMalloc(&out, 4);
  (void)hipMemset(src, 0, (size_t)rows * ld);
  suite<2, 2>(src, ld, ktiles, out);
  suite<4, 2>(src, ld, ktiles, out);
  suite<4, 4>(src, ld, ktiles, out);   // 256x256 block tile, 4 waves of 128x128 (256 accumulator registers), 1 block / CU
  return 0;
}

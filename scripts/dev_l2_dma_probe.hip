// dev probe (not part of the product): how fast can LDS-DMA pull GEMM-shaped operand tiles out of a warm L2, as a
// function of tiles in flight per block (D) and blocks per CU?  Tells whether the GEMM family's operand delivery is
// latency-bound (rate scales with bytes in flight) or throughput-bound (flat).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/dev_l2_dma_probe.hip -o scripts/_build/l2probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int D>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ src, long ld, int ktiles, int iters, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, b = blockIdx.x;
  // 256 rows per tile: rows 0..127 from "X panel" (b % 8), rows 128..255 from "W panel" 8 + (b / 8) % 8
  const long xrow = (long)(b % 8) * 128, wrow = (long)(8 + (b / 8) % 8) * 128;
  auto issue = [&](int t, int slot) {
    const int kt = t % ktiles;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int piece = j * 256 + tid;           // 2048 16-byte pieces: 256 rows x 8
      const int r = piece >> 3, c = piece & 7;
      const long row = r < 128 ? xrow + r : wrow + (r - 128);
      const char* g = src + row * ld + (long)kt * 128 + c * 16;
      char* l = lds + (size_t)slot * 32768 + (size_t)(j * 256 + (tid & ~63)) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    }
  };
#pragma unroll
  for (int d = 0; d < D; ++d) issue(d, d);
  for (int t = 0; t < iters; ++t) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * 8) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(t + D, t % D);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0 && lds[b & 1023] == 77) sink[0] = 1;
}

template <int D>
void run(const char* src, long ld, int ktiles, int blocks_per_cu, int* sink) {
  const int iters = 400;
  size_t lds = (size_t)D * 32768;
  const size_t cap = 160 * 1024 / blocks_per_cu;
  if (lds > cap) return;
  // pad the allocation so exactly blocks_per_cu blocks fit a CU
  size_t alloc = cap - 512;
  if (blocks_per_cu == 1 && alloc > 160 * 1024 - 1024) alloc = 160 * 1024 - 1024;
  hipFuncSetAttribute((const void*)probe<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)alloc);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  probe<D><<<grid, 256, alloc>>>(src, ld, ktiles, iters, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<D><<<grid, 256, alloc>>>(src, ld, ktiles, iters, sink);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)grid * (iters + D) * 32768.0;
  printf("D=%d blocks/CU=%d in-flight/CU=%3d KB  %7.2f TB/s  (%.1f GB/s per CU, %.2f us per tile per block)  err=%s\n", D,
         blocks_per_cu, D * blocks_per_cu * 32, bytes / ms / 1e9, bytes / ms / 1e6 / 256, ms * 1e3 / (iters + D),
         hipGetErrorString(hipGetLastError()));
}

int main() {
  const long ld = 2560;                 // K = 1280 bf16
  const int rows = 2048, ktiles = 20;
  char* src;
  int* sink;
  hipMalloc(&src, (size_t)rows * ld);
  hipMalloc(&sink, 4);
  hipMemset(src, 1, (size_t)rows * ld);
  for (int bpc : {1, 2, 4}) {
    run<1>(src, ld, ktiles, bpc, sink);
    run<2>(src, ld, ktiles, bpc, sink);
    run<3>(src, ld, ktiles, bpc, sink);
    run<4>(src, ld, ktiles, bpc, sink);
  }
  return 0;
}

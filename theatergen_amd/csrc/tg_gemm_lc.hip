// Long-K plain GEMM for the 320-multiple widths of the SD UNets: 128 x 320 output tiles, 4 COMPUTE + 4 LOADER waves per workgroup
// (include/theatergen_hip.h: tg_gemm mode 0, selected by the planner in tg_gemm.hip for K >= 1024: the FeedForward output
// projections net.2 (K = 4 C, models/attention.py:243-300) and the 16 x 16 level's attention / proj_in / proj_out projections).
//
// The structure is the slab conv kernel's (tg_conv_slab.hip) without the window: the 128x128 kernels of tg_gemm.hip top out near
// 0.55 PF on these shapes (every wave issues LDS-DMA and MFMA, operands for 2 * 128 * 128 * 64 FLOP are 32 KB), the 8-wave big
// tiles of tg_gemm_bt.hip lose the DMA issue time inside their MFMA streams.  Here:
//   * compute waves (one per SIMD, wave tile 64 x 160 = 2 x 5 MFMA tiles, 160 accumulators) issue ds_read_b128 + MFMA only: the
//     weight fragments roll (w[j] re-read for the next k-step right behind its two MFMAs), the token fragments alternate
//     between two sets; hand-counted lgkmcnt; ONE workgroup barrier per K-step (64 k = 4 k-steps x 10 MFMAs), before the last k-step;
//   * loader waves (one per SIMD) move the A tile (128 x 64, 16 instructions) and the W tile (320 x 64, 40 instructions) by
//     LDS-DMA into two 56 KB stages (128-byte rows, XOR swizzle on the source address), and run the NEXT work item's first two
//     K-steps under the epilogue (which bounces through its own 34 KB of LDS);
//   * persistent workgroups, XCD-chunked order; K split in 2 where 128 x 320 tiles alone would fill half the chip (M = 4096, N = 1280:
//     128 tiles): fp32 partial tiles + tg_gemm.hip's fixed-order reduce kernel (bias / residual / scale applied there).
#include "tg_gemm_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void lc_gemm_kernel(GemmParams p) {
  constexpr int BM = 128, BN = 320, TM = 2, TN = 5;
  constexpr unsigned A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE_BYTES = A_BYTES + W_BYTES, SCRATCH = 2 * STAGE_BYTES;
  constexpr int AJ = BM / 32, WJ = BN / 32;         // LDS-DMA instructions per loader wave per tile (8 rows x 128 B each)
  typedef typename Vec<T>::v8 V8;

  extern __shared__ __attribute__((aligned(128))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int lane = threadIdx.x & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const bool loader = wave8 >= 4;
  const int wave = wave8 & 3;

  const int S = p.tail_s;                            // K splits per tile (1 or 2 ...); work item w -> (tile w / S, split w % S)
  const int nkt_all = (int)(p.K / BK);
  const int kps = p.kt_per_split;                    // K-steps per split (the last split may be shorter)
  const int tiles_m = (int)(p.M / BM);
  const int nitems = tiles_m * p.tiles_n * S;

  if (loader) {
    const T* Ap = reinterpret_cast<const T*>(p.a0);
    const T* Wp = reinterpret_cast<const T*>(p.w);
    // instruction q = j * 4 + wave covers rows [8q, 8q + 8): lane -> (row 8q + lane / 8, slot lane % 8), chunk = slot ^ key(row),
    // key(row) = (row >> 1) & 7 = (4 (q & 1) + lane / 16) & 7
    const int chunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    auto dma = [&](const T* src, unsigned lds_byte_addr) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(src), "s"(lds_byte_addr)
                   : "memory");
    };
    const T* alane = nullptr;
    const T* wlane = nullptr;
    int k_begin = 0, nkt = 0;
    auto setup = [&](int w) {
      const int lbid = xcd_chunked_block_id(w, nitems);
      const int t = lbid / S, sp = lbid - t * S;
      const int tile_n = t % p.tiles_n, tile_m = t / p.tiles_n;
      const long m0 = (long)tile_m * BM, n0 = (long)tile_n * BN;
      k_begin = sp * kps;
      nkt = nkt_all - k_begin < kps ? nkt_all - k_begin : kps;
      alane = Ap + (m0 + wave * 8 + (lane >> 3)) * p.c0 + (long)k_begin * BK + chunk * 8;
      wlane = Wp + (n0 + wave * 8 + (lane >> 3)) * p.K + (long)k_begin * BK + chunk * 8;
    };
    auto issue = [&](int kt, int stage) {
      const unsigned dst = lds0 + (unsigned)stage * STAGE_BYTES + (unsigned)wave * 1024u;
      const long ko = (long)kt * BK;
#pragma unroll
      for (int j = 0; j < WJ; ++j) dma(wlane + ko + (long)j * 32 * p.K, dst + A_BYTES + (unsigned)j * 4096u);
#pragma unroll
      for (int j = 0; j < AJ; ++j) dma(alane + ko + (long)j * 32 * p.c0, dst + (unsigned)j * 4096u);
    };
    int w = blockIdx.x;
    if (w < nitems) {
      setup(w);
      issue(0, 0);
      if (nkt > 1) issue(1, 1);
    }
    for (; w < nitems; w += gridDim.x) {
      // item start: K-step 0 has landed (K-step 1 may be in flight)
      if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AJ + WJ) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      for (int kt = 0; kt < nkt; ++kt) {
        // seam of K-step kt: K-step kt + 1 (the only DMA in flight) has landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nkt) issue(kt + 2, kt & 1);     // every compute wave has its last fragments of this stage: refill it
      }
      const int wn = w + (int)gridDim.x;
      if (wn < nitems) {                             // the next item's first two K-steps, under the epilogue
        setup(wn);
        issue(0, 0);
        if (nkt > 1) issue(1, 1);
      }
    }
    return;
  }

  // ------------------------------------------------------------ compute waves ------------------------------------------------
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const unsigned sw = (((unsigned)hi ^ (unsigned)((l31 >> 1) & 7)) << 4);
  const unsigned fx0 = lds0 + (unsigned)((wave_m * TM * 32 + l31) * 128) + sw;
  const unsigned fw0 = lds0 + A_BYTES + (unsigned)((wave_n * TN * 32 + l31) * 128) + sw;
  unsigned ax, aw;
  auto set_stage = [&](int stage) {
    ax = fx0 + (unsigned)stage * STAGE_BYTES;
    aw = fw0 + (unsigned)stage * STAGE_BYTES;
  };
  auto read_x = [&](u32x4 (&xf)[TM], int ks) {
    const unsigned a = ax ^ ((unsigned)ks << 5);    // chunk 2 ks + hi: flips bits 5..6 of the swizzled slot (bases are 128-byte aligned)
    asm volatile("ds_read_b128 %0, %1" : "=v"(xf[0]) : "v"(a));
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(xf[1]) : "v"(a));
  };
  auto read_w = [&](u32x4& wf, int j, int ks) {
    const unsigned a = aw ^ ((unsigned)ks << 5);
    if (j == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(wf) : "v"(a));
    if (j == 1) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(wf) : "v"(a));
    if (j == 2) asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(wf) : "v"(a));
    if (j == 3) asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(wf) : "v"(a));
    if (j == 4) asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(wf) : "v"(a));
  };
  // one k-step: x_next <- token fragments of k-step `nks`, then per column j: wait, 2 MFMAs, re-read w[j] for k-step `nks`.
  // Reads issued after w[j] of k-step s and before its use: w[j+1..4] of s, x of s + 1, w[0..j-1] of s + 1 = 6 -> lgkmcnt(6).
  auto kstep = [&](f32x16 (&acc)[TM][TN], const u32x4 (&xc)[TM], u32x4 (&xn)[TM], u32x4 (&wf)[TN], int nks, bool have_next) {
    if (have_next) read_x(xn, nks);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      acc[0][j] = mfma32(__builtin_bit_cast(V8, wf[j]), __builtin_bit_cast(V8, xc[0]), acc[0][j]);
      acc[1][j] = mfma32(__builtin_bit_cast(V8, wf[j]), __builtin_bit_cast(V8, xc[1]), acc[1][j]);
      __builtin_amdgcn_sched_barrier(0);
      if (have_next) read_w(wf[j], j, nks);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  for (int w = blockIdx.x; w < nitems; w += gridDim.x) {
    const int lbid = xcd_chunked_block_id(w, nitems);
    const int t = lbid / S, sp = lbid - t * S;
    const int tile_n = t % p.tiles_n, tile_m = t / p.tiles_n;
    const long m0 = (long)tile_m * BM, n0 = (long)tile_n * BN;
    const int k_begin = sp * kps;
    const int nkt = nkt_all - k_begin < kps ? nkt_all - k_begin : kps;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    __builtin_amdgcn_s_barrier();                   // item start (see the loader)
    __builtin_amdgcn_sched_barrier(0);
    u32x4 xa[TM], xb[TM], wf[TN];
    set_stage(0);
    read_x(xa, 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) read_w(wf[j], j, 0);
    for (int kt = 0; kt < nkt; ++kt) {
      kstep(acc, xa, xb, wf, 1, true);
      kstep(acc, xb, xa, wf, 2, true);
      kstep(acc, xa, xb, wf, 3, true);
      // seam: all my reads of this stage were issued above: wait for them, then the workgroup barrier; behind it the loaders
      // refill the stage and the last k-step reads the NEXT K-step's fragments
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const bool last = kt + 1 == nkt;
      if (!last) set_stage((kt + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      kstep(acc, xb, xa, wf, 0, !last);
    }
    // every wave is past the last barrier with all its fragment reads done; the next item's prologue leaves the bounce region alone
    epilogue_tile_lds<T, TM, TN, 0>(p, acc, m0 + wave_m * TM * 32, n0 + wave_n * TN * 32, lane,
                                   reinterpret_cast<float*>(smem + SCRATCH) + wave * (32 * 68), S > 1 ? lbid : -1, m0, n0);
  }
}

template <typename T>
int launch_lc(const tg_gemm_desc* d, GemmParams p, int splits, hipStream_t st) {
  const size_t lds = 2 * (size_t)(128 + 320) * 128 + (size_t)4 * 32 * 68 * 4;
  const long tiles_m = d->M / 128, tiles_n = d->N / 320;
  const int nkt = (int)(d->K / BK);
  p.tiles_n = (int)tiles_n;
  p.full_tiles = 0;                               // every tile is a "tail" tile of `splits` K ranges for the reduce kernel
  p.tail_s = splits;
  p.kt_per_split = (nkt + splits - 1) / splits;
  p.tile_bm = 128; p.tile_bn = 320;
  long grid = tiles_m * tiles_n * splits;
  if (grid > 256) grid = 256;                     // one persistent workgroup per CU
  auto k = lc_gemm_kernel<T>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr;
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(512), lds, st, p);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

}  // namespace

// Called by tg_gemm.hip's planner (not part of the C ABI); GemmParams arrives filled except for the tile bookkeeping.  With
// splits > 1 the caller runs the reduce kernel over tiles_m * tiles_n tail tiles of `splits` partials each.
int tg_gemm_lc_launch(const tg_gemm_desc* d, const void* params, int splits, void* stream) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == TG_BF16) return launch_lc<bf16_t>(d, p, splits, st);
  return launch_lc<f16_t>(d, p, splits, st);
}

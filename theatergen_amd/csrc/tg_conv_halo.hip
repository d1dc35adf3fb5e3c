// conv_halo_kernel: conv3x3 (stride 1, pad 1) with an LDS-staged halo window; launched from tg_gemm.hip through
// tg_conv_halo_launch (own translation unit: compile time).
#include "tg_gemm_common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------
// conv3x3 with an LDS-STAGED HALO WINDOW (stride 1, pad 1, image width 16 / 32 / 64).
// The implicit-GEMM kernel above fetches every activation row 9 times (once per tap) through L2; the GEMM family is
// bound by L2 -> LDS operand delivery (profiles/r1_gemm_findings.md), so here a block = 128 output pixels = TH full
// image rows stages the (TH+2) x (W+2) input halo of ONE 64-channel chunk in LDS once and serves all 9 taps from
// it: the MFMA B-operand (lane = pixel) is read at slab row (py+ky)*(W+2) + px+kx.  K runs chunk-major / tap-minor;
// only the 128x64 weight tile streams per K-step (double-buffered).  Activation L2 traffic drops 9x, total operand
// traffic per FLOP by ~1.7x.  Same swizzle, same accumulator layout and the same epilogue as the GEMM kernel
// (full-width rows make the block's pixels contiguous in the token-major tensor).
// (Round 5 built a 128 x 160-channel instance for the 8 x 8 level — 64 tiles x 4 K splits = exactly one workgroup per CU, double-buffered window, five-stage
// weight ring, then eight waves sharing each K-step — parity-tested, and measured it at 55.2-57.8 us against this kernel's 56.9 (K = 11520) and SLOWER on the
// two-source layers (93 vs 86 us): a K-step takes ~0.9-1.0 us whatever the structure; removed.  profiles/r5_t160_findings.md.)
// UPS = true: the same for Upsample2D (nearest x2 then conv3x3): WI is the OUTPUT width, the slab holds the
// (TH/2 + 2) x (WI/2 + 2) INPUT pixels the block's upsampled window maps to (input pixel = upsampled coordinate >> 1).
template <typename T, int WI, bool UPS>
__global__ __launch_bounds__(256) void conv_halo_kernel(GemmParams p) {
  constexpr int BM = 128, BN = 128, NW = 4, TM = 2, TN = 2, WAVES_N = 2;
  // WI = 8 (the 8x8 level): a block's 128 pixels are TWO whole 8x8 images; their two 10x10 padded windows are stacked
  // in the slab (20 slab rows of width 10), everything else is unchanged
  constexpr bool MULTI = WI == 8;
  static_assert(!(MULTI && UPS), "no upsample variant at width 8");
  constexpr int TH = BM / WI, WIN = UPS ? WI / 2 : WI, SW = WIN + 2, SROWS = MULTI ? 20 : (UPS ? TH / 2 + 2 : TH + 2);
  constexpr int SLAB = SROWS * SW, NI = (SLAB + 7) / 8, SJ = (NI + NW - 1) / NW;
  constexpr int WJ = BN / (8 * NW);
  typedef typename Vec<T>::v8 V8;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sS = reinterpret_cast<T*>(smem);                 // [NI*8][64]   halo slab of the current channel chunk
  stagger_first_round(p.flags, smem);
  // weight-tile ring: 3 stages (2 tiles in flight) wherever slab + 3 x 16 KB still lets two blocks share a CU (every
  // variant but the 64-wide one): a K-step's period was one DMA round trip of the next W tile, not its 16 MFMAs
  constexpr int WST = ((size_t)NI * 8 * BK + 3 * BN * BK) * sizeof(T) <= 80 * 1024 ? 3 : 2;
  constexpr int WD = WST - 1;                          // W tiles issued ahead of the one being multiplied
  T* sW = sS + NI * 8 * BK;                           // [WST][BN][64]  weight tiles

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / WAVES_N;
  const int wave_n = wave % WAVES_N;
  int lbid, split = 0, part = -1;     // same work-item scheme as gemm_glds_kernel; the K split runs over channel chunks
  if ((int)blockIdx.x < p.full_tiles) {
    lbid = xcd_chunked_block_id(blockIdx.x, p.full_tiles);
  } else {
    const int j = (int)blockIdx.x - p.full_tiles;
    lbid = p.full_tiles + j / p.tail_s;
    split = j - (j / p.tail_s) * p.tail_s;
    part = j;
  }
  const int tile_n = lbid % p.tiles_n;
  const int tile_m = lbid / p.tiles_n;
  const long m0 = (long)tile_m * BM;
  const long n0 = (long)tile_n * BN;
  const int H = p.in_h;                               // INPUT height (output height = 2H when UPS)
  const int HO = UPS ? 2 * H : H;
  const int img = (int)(m0 / ((long)HO * WI));
  const int y0 = (int)((m0 - (long)img * HO * WI) / WI);
  const int iy0 = UPS ? y0 / 2 - 1 : y0 - 1;          // input row held by slab row 0

  const int lrow = lane >> 3;
  const int slot = lane & 7;
  const int wkey = (4 * (wave & 1) + (lane >> 4)) & 7;
  const int chunk = slot ^ wkey;

  const T* A0 = reinterpret_cast<const T*>(p.a0);
  const T* A1 = reinterpret_cast<const T*>(p.a1);
  const T* Wp = reinterpret_cast<const T*>(p.w);
  const T* zero = reinterpret_cast<const T*>(tg_zero_page);
  const int ctot = p.c0 + p.c1;
  const int nchunks = ctot / BK;

  int spix[SJ];                                       // input pixel feeding this lane's slab row (-1: zero padding)
#pragma unroll
  for (int j = 0; j < SJ; ++j) {
    const int sr = (j * NW + wave) * 8 + lrow;
    const int sy = sr / SW, sx = sr - sy * SW;
    int iy = iy0 + sy, im = img;
    if (MULTI) { im = img + sy / 10; iy = sy % 10 - 1; }       // slab rows [10 i, 10 i + 10) = padded window of image img + i
    const int ix = sx - 1;
    const bool ok = sr < SLAB && iy >= 0 && iy < H && ix >= 0 && ix < WIN && (m0 < p.M);
    spix[j] = ok ? (im * H + iy) * WIN + ix : -1;
  }
  const T* wrow[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) {
    const long n = n0 + (j * NW + wave) * 8 + lrow;
    wrow[j] = n < p.N ? Wp + n * p.K : nullptr;
  }

  auto dma = [&](const T* src, T* lds_row_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_row_base, 16, 0, 0);
  };
  auto issue_slab = [&](int cc) {
    int c = cc * BK;
    const T* base = A0;
    int pitch = p.c0;
    if (c >= p.c0) { base = A1; pitch = p.c1; c -= p.c0; }
    c += chunk * 8;
#pragma unroll
    for (int j = 0; j < SJ; ++j) {
      if (j * NW + wave < NI) {
        const T* src = spix[j] >= 0 ? base + (long)spix[j] * pitch + c : zero;
        dma(src, sS + (j * NW + wave) * 8 * BK);
      }
    }
  };
  // weight tile requests of a column tile wholly inside N: wave-uniform base (weight pointer + the K-step's column offset) + this lane's constant 32-bit byte offset
  // (tg_gemm_glds.h, round 6: no 64-bit pointer arithmetic / zero-page select per request in the K loop)
  const unsigned lds_sw = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)sW;
  const bool fastw = n0 + BN <= p.N && (long)p.N * p.K * (long)sizeof(T) < (1L << 31);
  unsigned voffw[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) voffw[j] = (unsigned)(((n0 + (j * NW + wave) * 8 + lrow) * p.K + chunk * 8) * (long)sizeof(T));
  auto dma_s = [&](const T* base, unsigned voff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_byte_addr) : "memory");
  };
  auto issue_w = [&](int cc, int tap, int buf) {
    if (fastw) {
      const T* wb = Wp + ((long)tap * ctot + cc * BK);
#pragma unroll
      for (int j = 0; j < WJ; ++j) dma_s(wb, voffw[j], lds_sw + (unsigned)((buf * BN * BK + (j * NW + wave) * 8 * BK) * sizeof(T)));
      return;
    }
    const long kc = (long)tap * ctot + cc * BK + chunk * 8;
    T* dw = sW + buf * BN * BK;
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const T* src = wrow[j] != nullptr ? wrow[j] + kc : zero;
      dma(src, dw + (j * NW + wave) * 8 * BK);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31;
  const int hi = lane >> 5;
  const int rkey = (l31 >> 1) & 7;
  int ppy[TM], ppx[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int pm = wave_m * TM * 32 + i * 32 + l31;
    ppy[i] = MULTI ? (pm >> 6) * 10 + ((pm & 63) >> 3) : pm / WI;
    ppx[i] = pm % WI;
  }

  // channel chunks of this work item (kt_per_split counts chunks here)
  const int c_begin = part >= 0 ? split * p.kt_per_split : 0;
  int c_end = part >= 0 ? c_begin + p.kt_per_split : nchunks;
  if (c_end > nchunks) c_end = nchunks;

  const int nkt = (c_end - c_begin) * 9;
  int icc = c_begin, itap = 0;                         // (chunk, tap) of the next W tile to request
  issue_slab(c_begin);
#pragma unroll
  for (int s_ = 0; s_ < WD; ++s_) {
    if (s_ < nkt) {
      issue_w(icc, itap, s_);
      if (++itap == 9) { itap = 0; ++icc; }
    }
  }
  if (WD == 2 && nkt >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WJ) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int cc = c_begin, tap = 0;
  int buf = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const T* bw = sW + buf * BN * BK + (wave_n * TN * 32 + l31) * BK;
    V8 xf[BK / 16][TM], wf[BK / 16][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int sr = UPS ? (((ppy[i] + ky - 1) >> 1) + 1) * SW + ((ppx[i] + kx - 1) >> 1) + 1
                         : (ppy[i] + ky) * SW + ppx[i] + kx;
      const int key = (sr >> 1) & 7;
      const T* bx = sS + sr * BK;
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) xf[ks][i] = *reinterpret_cast<const V8*>(bx + ((2 * ks + hi) ^ key) * 8);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int so = ((2 * ks + hi) ^ rkey) * 8;
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[ks][j] = *reinterpret_cast<const V8*>(bw + j * 32 * BK + so);
    }
    __builtin_amdgcn_sched_barrier(0);
    int ncc = cc, ntap = tap + 1;
    if (ntap == 9) { ntap = 0; ncc = cc + 1; }
    if (kt + 1 < nkt) {
      if (ntap == 0) {
        // the next K-step starts a new channel chunk: every wave must have its tap-8 fragments in registers before
        // the slab is overwritten; the slab DMA then overlaps this step's 16 MFMAs
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue_slab(ncc);
      }
    }
    if (kt + WD < nkt) {
      int nb = buf + WD;
      if (nb >= WST) nb -= WST;
      issue_w(icc, itap, nb);
      if (++itap == 9) { itap = 0; ++icc; }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(wf[ks][j], xf[ks][i], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
    // the W tile of step kt+1 (and a slab requested in this step, which is older than this step's W request) must have
    // landed; the W tile requested in this step may stay in flight
    if (WD == 2 && kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WJ) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    cc = ncc;
    tap = ntap;
    buf = buf + 1 == WST ? 0 : buf + 1;
  }

  epilogue_tile_lds<T, TM, TN, 0>(p, acc, m0 + wave_m * TM * 32, n0 + wave_n * TN * 32, lane,
                                 reinterpret_cast<float*>(smem) + wave * (32 * (TN * 32 + 4)), part, m0, n0);
}

template <typename T, int WI, bool UPS>
int launch_halo(const GemmParams& p, int grid, hipStream_t st) {
  constexpr int TH = 128 / WI, SLAB = WI == 8 ? 200 : (UPS ? (TH / 2 + 2) * (WI / 2 + 2) : (TH + 2) * (WI + 2)), NI = (SLAB + 7) / 8;
  const int wst = ((size_t)NI * 8 * BK + 3 * 128 * BK) * sizeof(T) <= 80 * 1024 ? 3 : 2;
  const size_t lds = ((size_t)NI * 8 * BK + wst * 128 * BK) * sizeof(T);
  auto k = conv_halo_kernel<T, WI, UPS>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr;
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, st, p);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

template <typename T>
int dispatch_halo(const tg_gemm_desc* d, const GemmParams& p, int grid, hipStream_t st) {
  if (d->upsample) {
    if (d->out_w == 64) return launch_halo<T, 64, true>(p, grid, st);
    if (d->out_w == 32) return launch_halo<T, 32, true>(p, grid, st);
    return launch_halo<T, 16, true>(p, grid, st);
  }
  if (d->out_w == 64) return launch_halo<T, 64, false>(p, grid, st);
  if (d->out_w == 32) return launch_halo<T, 32, false>(p, grid, st);
  if (d->out_w == 8) return launch_halo<T, 8, false>(p, grid, st);
  return launch_halo<T, 16, false>(p, grid, st);
}

}  // namespace

// grid = full tiles + tail tiles * K splits (the caller, tg_gemm.hip, owns the plan and launches the split reduce)
int tg_conv_halo_launch(const tg_gemm_desc* d, const void* params, int grid, void* stream) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == TG_BF16) return dispatch_halo<bf16_t>(d, p, grid, st);
  return dispatch_halo<f16_t>(d, p, grid, st);
}

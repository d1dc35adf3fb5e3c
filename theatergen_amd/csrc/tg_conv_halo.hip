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
// UPS = true: the same for Upsample2D (nearest x2 then conv3x3): WI is the OUTPUT width, the slab holds the
// (TH/2 + 2) x (WI/2 + 2) INPUT pixels the block's upsampled window maps to (input pixel = upsampled coordinate >> 1).
// BNT = 160 (round 5; the 8 x 8 level only): 128 pixels x 160 channels per workgroup, four waves of 32 pixels x 160 channels.  At CFG batch 16 the level
// is M = 1024 pixels: 8 x 8 = 64 tiles x 4 K splits = 256 work items = exactly ONE per CU (128 x 128: 80 tiles x 5 splits = 400 items on 512 slots, 144 CUs
// with two, 112 with one).  One workgroup per CU owns the whole LDS: the slab is DOUBLE-buffered (the next chunk's window is requested five K-steps before it
// is needed, no barrier / drain at the chunk boundary) and the weight ring is five 20 KB stages deep (four tiles in flight: a K-step's period was one DMA
// round trip).  EIGHT waves: two groups of four share every K-step — group g multiplies k-steps 2g, 2g + 1 of the 64-wide K-tile on the same 32 x 160 wave
// tiles (12 fragment reads + 10 MFMAs per wave and step instead of 24 + 20, W requests alternate between the groups): a lone wave per SIMD pays its LDS
// reads, its DMA issue and its MFMAs as a SUM (first version, four waves: 57.8 us = the 128 x 128 instance's 57.3), two waves per SIMD overlap them.  The
// groups' partial accumulators meet in LDS after the loop (fixed order: group 0 + group 1), group 0 runs the epilogue.
template <int N> __device__ __forceinline__ void halo_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T, int WI, bool UPS, int BNT = 128>
__global__ __launch_bounds__(BNT == 128 ? 256 : 512) void conv_halo_kernel(GemmParams p) {
  constexpr int BM = 128, BN = BNT, NW = 4, WAVES_N = BN == 128 ? 2 : 1, WAVES_M = NW / WAVES_N, TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
  constexpr bool DEEP = BN != 128;
  static_assert(!DEEP || (WI == 8 && !UPS && BN == 160), "the deep-ring instance is the 8 x 8 level's");
  // WI = 8 (the 8x8 level): a block's 128 pixels are TWO whole 8x8 images; their two 10x10 padded windows are stacked
  // in the slab (20 slab rows of width 10), everything else is unchanged
  constexpr bool MULTI = WI == 8;
  static_assert(!(MULTI && UPS), "no upsample variant at width 8");
  constexpr int TH = BM / WI, WIN = UPS ? WI / 2 : WI, SW = WIN + 2, SROWS = MULTI ? 20 : (UPS ? TH / 2 + 2 : TH + 2);
  constexpr int NWD = DEEP ? 2 * NW : NW;           // waves that issue the slab's DMA (all of them)
  constexpr int NG = DEEP ? 2 : 1, KSG = (BK / 16) / NG;     // wave groups, k-steps of a K-tile per group
  constexpr int SLAB = SROWS * SW, NI = (SLAB + 7) / 8, SJ = (NI + NWD - 1) / NWD;
  constexpr int WJ = BN / (8 * NW);
  typedef typename Vec<T>::v8 V8;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sS = reinterpret_cast<T*>(smem);                 // [NI*8][64]   halo slab of the current channel chunk
  stagger_first_round(p.flags, smem);
  // weight-tile ring: 3 stages (2 tiles in flight) wherever slab + 3 x 16 KB still lets two blocks share a CU (every
  // variant but the 64-wide one): a K-step's period was one DMA round trip of the next W tile, not its 16 MFMAs
  constexpr int WST = DEEP ? 5 : (((size_t)NI * 8 * BK + 3 * BN * BK) * sizeof(T) <= 80 * 1024 ? 3 : 2);
  constexpr int WD = WST - 1;                          // W tiles issued ahead of the one being multiplied
  constexpr int NSLAB = DEEP ? 2 : 1;
  T* sW = sS + NSLAB * NI * 8 * BK;                   // [WST][BN][64]  weight tiles

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = DEEP ? wave_all >> 2 : 0;           // DEEP: which half of every K-tile's k-steps this wave multiplies
  const int wave = wave_all & 3;
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
    const int sr = (j * NWD + wave_all) * 8 + lrow;
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
  // channel chunks of this work item (kt_per_split counts chunks here)
  const int c_begin = part >= 0 ? split * p.kt_per_split : 0;
  int c_end = part >= 0 ? c_begin + p.kt_per_split : nchunks;
  if (c_end > nchunks) c_end = nchunks;
  // CHUNK ROTATION (round 5, p.k_rot): work item lbid walks its chunks starting at chunk lbid % (number of chunks) and wraps — the workgroups of a launch run in
  // lockstep and would otherwise all ask the L2 for the same K offset (same few channels) at the same time; see tg_gemm_glds.h.  cc below = LOGICAL chunk.
  const int ncl = c_end - c_begin;
  const int crot = (p.k_rot != 0 && ncl > 1) ? lbid % ncl : 0;
  auto pchunk = [&](int cc) { int t = cc - c_begin + crot; if (t >= ncl) t -= ncl; return c_begin + t; };
  auto issue_slab = [&](int ccl, int sb = 0) {
    const int cc = pchunk(ccl);
    int c = cc * BK;
    const T* base = A0;
    int pitch = p.c0;
    if (c >= p.c0) { base = A1; pitch = p.c1; c -= p.c0; }
    c += chunk * 8;
#pragma unroll
    for (int j = 0; j < SJ; ++j) {
      if (j * NWD + wave_all < NI) {
        const T* src = spix[j] >= 0 ? base + (long)spix[j] * pitch + c : zero;
        dma(src, sS + sb * NI * 8 * BK + (j * NWD + wave_all) * 8 * BK);
      }
    }
  };
  auto issue_w = [&](int ccl, int tap, int buf) {
    const int cc = pchunk(ccl);
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

  const int nkt = (c_end - c_begin) * 9;
  int icc = c_begin, itap = 0;                         // (chunk, tap) of the next W tile to request
  issue_slab(c_begin);
#pragma unroll
  for (int s_ = 0; s_ < WD; ++s_) {
    if (s_ < nkt) {
      if (!DEEP || (s_ & 1) == grp) issue_w(icc, itap, s_);      // DEEP: W tile t is requested (and later awaited) by group t & 1
      if (++itap == 9) { itap = 0; ++icc; }
    }
  }
  if constexpr (DEEP) {
    // own requests so far: slab, W(grp), W(grp + 2).  Slab + W(0) landed: group 0 lets its younger W(2) fly, group 1 (W(1), W(3) both younger than the slab)
    // lets both fly; nkt >= 9
    if (grp == 0) halo_wait_vm<WJ>(); else halo_wait_vm<2 * WJ>();
  } else {
    if (WD == 2 && nkt >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WJ) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  int cc = c_begin, tap = 0;
  int buf = 0, cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const T* bw = sW + buf * BN * BK + (wave_n * TN * 32 + l31) * BK;
    const T* sSc = sS + cur * NI * 8 * BK;
    V8 xf[KSG][TM], wf[KSG][TN];
    const int ks0 = KSG * grp;                          // first k-step of this wave's group
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int sr = UPS ? (((ppy[i] + ky - 1) >> 1) + 1) * SW + ((ppx[i] + kx - 1) >> 1) + 1
                         : (ppy[i] + ky) * SW + ppx[i] + kx;
      const int key = (sr >> 1) & 7;
      const T* bx = sSc + sr * BK;
#pragma unroll
      for (int ks = 0; ks < KSG; ++ks) xf[ks][i] = *reinterpret_cast<const V8*>(bx + ((2 * (ks0 + ks) + hi) ^ key) * 8);
    }
#pragma unroll
    for (int ks = 0; ks < KSG; ++ks) {
      const int so = ((2 * (ks0 + ks) + hi) ^ rkey) * 8;
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[ks][j] = *reinterpret_cast<const V8*>(bw + j * 32 * BK + so);
    }
    __builtin_amdgcn_sched_barrier(0);
    int ncc = cc, ntap = tap + 1;
    if (ntap == 9) { ntap = 0; ncc = cc + 1; }
    if constexpr (DEEP) {
      // next chunk's window into the OTHER slab buffer, five K-steps ahead of its first use (last read a whole chunk ago: no barrier needed);
      // issued BEFORE this step's W tile, so it is older than the W tile whose landing the chunk's first step waits for
      if (tap == 3 && cc + 1 < c_end) issue_slab(cc + 1, cur ^ 1);
    } else {
      if (kt + 1 < nkt) {
        if (ntap == 0) {
          // the next K-step starts a new channel chunk: every wave must have its tap-8 fragments in registers before
          // the slab is overwritten; the slab DMA then overlaps this step's 16 MFMAs
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          issue_slab(ncc);
        }
      }
    }
    if (kt + WD < nkt) {
      int nb = buf + WD;
      if (nb >= WST) nb -= WST;
      if (!DEEP || ((kt + WD) & 1) == grp) issue_w(icc, itap, nb);
      if (++itap == 9) { itap = 0; ++icc; }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KSG; ++ks)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(wf[ks][j], xf[ks][i], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
    // the W tile of step kt+1 (and a slab requested in this step, which is older than this step's W request) must have
    // landed; the W tile requested in this step may stay in flight
    if constexpr (DEEP) {
      // W tiles younger than W(kt + 1) that may stay in flight: W(kt + 2) .. W(min(kt + WD, nkt - 1)).  (While a slab request is younger than
      // W(kt + 1) — three steps per chunk — the count under-states what is in flight and the wait retires a little more than it has to.)
      // DEEP: W(kt + 1) was requested by group (kt + 1) & 1 — its waves wait for it, letting their one younger request W(kt + 3) fly; the other
      // group's next tile W(kt + 2) is awaited a step later.  (A slab request younger than the awaited tile makes the wait retire a little more.)
      if (((kt + 1) & 1) == grp) {
        if (kt + 3 < nkt) halo_wait_vm<WJ>(); else halo_wait_vm<0>();
      }
    } else {
      if (WD == 2 && kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WJ) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    cc = ncc;
    tap = ntap;
    if (DEEP && ntap == 0) cur ^= 1;
    buf = buf + 1 == WST ? 0 : buf + 1;
  }

  constexpr int SCW = TN <= 2 ? TN : 2;
  float* scr = reinterpret_cast<float*>(smem) + wave * (32 * (SCW * 32 + 4));
  if constexpr (DEEP) {
    // the two groups' partial sums of the same wave tile meet in LDS (operand stages are dead: every request was awaited above): group 1 parks its 80
    // accumulator registers per lane, group 0 adds them (fp32, fixed order) and runs the epilogue with its scratch BEHIND the parked data
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x4* park = reinterpret_cast<f32x4*>(smem) + (wave * TM * TN * 4) * 64 + lane;
    if (grp == 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            park[((i * TN + j) * 4 + g) * 64] = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) return;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 o = park[((i * TN + j) * 4 + g) * 64];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] += o[e];
        }
    scr = reinterpret_cast<float*>(smem) + NW * TM * TN * 4 * 64 * 4 + wave * (32 * (SCW * 32 + 4));
  }
  epilogue_tile_lds<T, TM, TN, 0>(p, acc, m0 + wave_m * TM * 32, n0 + wave_n * TN * 32, lane, scr, part, m0, n0);
}

template <typename T, int WI, bool UPS, int BNT = 128>
int launch_halo(const GemmParams& p, int grid, hipStream_t st) {
  constexpr int TH = 128 / WI, SLAB = WI == 8 ? 200 : (UPS ? (TH / 2 + 2) * (WI / 2 + 2) : (TH + 2) * (WI + 2)), NI = (SLAB + 7) / 8;
  const int wst = BNT != 128 ? 5 : (((size_t)NI * 8 * BK + 3 * 128 * BK) * sizeof(T) <= 80 * 1024 ? 3 : 2);
  const size_t lds = ((size_t)(BNT != 128 ? 2 : 1) * NI * 8 * BK + wst * BNT * BK) * sizeof(T);
  auto k = conv_halo_kernel<T, WI, UPS, BNT>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr;
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(BNT == 128 ? 256 : 512), lds, st, p);
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
  if (d->out_w == 8) return p.tile_bn == 160 ? launch_halo<T, 8, false, 160>(p, grid, st) : launch_halo<T, 8, false>(p, grid, st);
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

// 3x3 convolution for the 320-multiple channel counts of the SD UNets: one output tile = 128 pixels x 320 output channels per
// workgroup, the input window staged ONCE per 320 channels, GroupNorm(+SiLU) applied while it is staged
// (include/theatergen_hip.h: tg_gemm mode 1, selected by the planner in tg_gemm.hip; replaces the conv1 / conv2 +
// nonlinearity(norm(x)) pairs of ResnetBlock2D, models/resnet.py via models/unet_2d_blocks.py:184-195).
//
// Why a second conv kernel.  conv_halo_kernel (tg_gemm.hip) gives every 128-channel output tile its own workgroup: N = 320
// costs three tiles (the third half empty: 17 % of the MFMA work wasted), every tile re-stages the input window (measured
// 272 MB per launch against 86 MB algorithmic on the 64^2 320 -> 320 layer), the window of the next channel chunk can only be
// requested after the last read of the current one (single slab buffer), and because the slab goes HBM -> LDS by DMA nothing
// can be applied to it on the way: GroupNorm + SiLU was a separate pass that wrote the normalised tensor to HBM and read it
// back.  Here a workgroup is 4 COMPUTE waves + 4 LOADER waves (one of each per SIMD):
//   * compute waves: wave tile 64 x 160 = 2 x 5 MFMA tiles of 32 x 32 (160 accumulator registers); 7 fragment reads feed 10
//     MFMAs per k-step (LDS read time 1/3 of MFMA time); fragments double-buffered in registers by inline-asm ds_read_b128 with
//     hand-counted lgkmcnt; they issue NO memory instruction in the K loop.  Measured on the first build of this kernel (4 waves
//     doing everything, scripts/dev_slab_exp.py): without the ten weight LDS-DMA instructions per K-step in the MFMA wave's
//     instruction stream the 64^2 960 -> 320 layer ran in 273 us instead of 382 — an LDS-DMA instruction occupies its wave's
//     issue for 60-180 cycles (MI355X_MICROARCH.md), and with one wave per SIMD that is matrix-pipe idle time;
//   * loader waves: weight tiles (320 x 64, 40 KB) by LDS-DMA into a ring of THREE stages (a tile is requested two K-steps =
//     2560 matrix-pipe cycles before its first read: with two stages the barrier waited for the DMA round trip); the input
//     window (slab: (128 / W + 2) x (W + 2) pixels x 64 channels, 128-byte swizzled rows) goes global -> registers -> LDS:
//     requested at tap 0 of the previous chunk, normalised in the registers (x * a[b, c] + d[b, c], SiLU — the same fp32
//     expression and rounding point as tg_groupnorm's apply pass, so the bf16 / fp16 MFMA inputs are bit-identical to the
//     unfused path) a row or two per K-step by the loaders' VALU while the compute waves keep the matrix pipe busy, written at
//     the chunk boundary; after the last K-step of a tile they run the NEXT tile's prologue under the epilogue;
//   * a K-step is 4 k-steps x 10 MFMAs = 1280 matrix-pipe cycles; ONE workgroup barrier per K-step, placed before the last k-step
//     (as in tg_gemm_bt.hip): behind it the stage just read is refilled and the first fragments of the next K-step are read;
//   * persistent workgroups, one per CU, XCD-chunked tile order; epilogue = the shared LDS-transposed one (bias + time-embedding
//     vector + residual + scale, 16-byte stores on whole 128-byte rows) bouncing through the slab region.
// K order (channel chunk, tap, k) is conv_halo_kernel's: the two kernels produce bit-identical results on the same input.
#include "tg_gemm_common.h"

namespace {

// WI = patch (or image) width, NP = patches per tile: NP = 1: one TH x WI patch, TH = 128 / WI rows (whole image rows when in_w == WI);
// NP = 2 (WI = 8): two 8 x 8 patches, their 10 x 10 windows stacked in the slab — the 8 x 8 level (two whole images per tile) and
// the 24- / 40-wide maps; one compute-wave row (64 pixels) is one patch.
// PATCH = false: whole-row tiles (in_w == WI, NP = 1): the instances of the SD-1.5 bench, kept free of the patch code paths.
template <typename T, int WI, int NP, bool PRO, bool PATCH>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_slab_kernel(GemmParams p) {
  constexpr int BM = 128, BN = 320, TM = 2, TN = 5, NF = TM + TN;
  constexpr int PP = BM / NP, TH = PP / WI, SW = WI + 2, SROWS = TH + 2, WIN = SROWS * SW, SLAB = NP * WIN, SJ = (SLAB + 31) / 32;
  static_assert(NP == 1 || (NP == 2 && PP == 64), "two patches per tile = one per compute-wave row");
  static_assert(PATCH || NP == 1, "two-patch tiles are patch tiles");
  constexpr unsigned SLAB_BYTES = SJ * 32 * 128, WST_BYTES = BN * 128, SCRATCH_BYTES = 4 * 32 * 68 * 4;
  // LDS: slab (ONE buffer; the epilogue bounce lives here too) | weight stages 0, 1, 2.  Three 40 KB weight stages leave room for
  // one slab only: the loaders hold the NEXT chunk's window in registers (loaded and normalised under the current chunk) and
  // write it between the last K-step of a chunk and the first of the next (one extra barrier per 9 K-steps).
  constexpr unsigned SLAB_PAD = SLAB_BYTES > SCRATCH_BYTES ? SLAB_BYTES : SCRATCH_BYTES, W_BASE = SLAB_PAD;
  constexpr int WJ = BN / 32;                      // LDS-DMA instructions per loader wave per weight tile (8 rows x 128 B each)
  static_assert(PP % WI == 0, "whole image (or patch) rows per tile");
  static_assert(SJ <= 9, "two slab rows per thread in taps 3..5, one in taps 6..8");
  typedef typename Vec<T>::v8 V8;

  extern __shared__ __attribute__((aligned(128))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int lane = threadIdx.x & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const bool loader = wave8 >= 4;
  const int wave = wave8 & 3;                      // index within the role
  const int tid = (int)threadIdx.x & 255;          // thread index within the role

  const int ctot = p.c0 + p.c1;
  const int nchunks_all = ctot / BK;
  // K split over channel chunks (the 16-wide maps: M = 4096 is 32 row tiles x 4 column tiles = half the chip): work item
  // w -> (tile w / S, split w % S), split sp owns chunks [sp * cps, min((sp + 1) * cps, all)), fp32 partial tiles + tg_gemm.hip's
  // fixed-order reduce kernel (bias / vector / residual / scale applied there)
  const int S = p.tail_s, cps = p.kt_per_split;
  const int H = p.in_h;
  const int tiles_m = (int)(p.M / BM);
  const int ntiles = tiles_m * p.tiles_n * S;       // work items

  // work item v -> (tile t = tile_m * tiles_n + tile_n, split sp).  XCD-chunked order in both forms: XCD x owns a contiguous range of
  // logical ids.  Default: tile-major (an XCD's range is a band of row tiles over all column tiles and splits: every XCD streams the
  // WHOLE weight tensor through its L2, the window of a row tile is shared).  Weight-heavy layers (p.slab_order = 1: weight bytes >
  // activation bytes, the 16-wide and smaller maps): (column tile, split)-major / row-tile-minor — an XCD's range covers all row
  // tiles of one or two (column tile, split) weight slices, e.g. SD-1.5's 16 x 16 level: 32 row tiles x 8 slices of 3.7 MB = one
  // L2-sized slice per XCD, and the weights cross the fabric once instead of eight times.
  auto work_item = [&](int v, int& t, int& sp) {
    const int lbid = xcd_chunked_block_id(v, ntiles);
    if (p.slab_order == 1) {
      const int combo = lbid / tiles_m, tm_ = lbid - combo * tiles_m;
      const int tn_ = combo / S;
      sp = combo - tn_ * S;
      t = tm_ * p.tiles_n + tn_;
    } else {
      t = lbid / S;
      sp = lbid - t * S;
    }
  };

  if (loader) {
    // =========================================================== loader waves ===========================================

    const T* A0 = reinterpret_cast<const T*>(p.a0);
    const T* A1 = reinterpret_cast<const T*>(p.a1);
    const T* Wp = reinterpret_cast<const T*>(p.w);
    const T* zero = reinterpret_cast<const T*>(tg_zero_page);
    const float* coef = reinterpret_cast<const float*>(p.a_coef);
    // weight tiles: instruction q = j * 4 + wave covers rows [8q, 8q + 8): lane -> (row 8q + lane / 8, slot lane % 8), chunk
    // fetched into a slot = slot ^ key(row), key(row) = (row >> 1) & 7 (conflict-free ds_read_b128 of 32 consecutive rows)
    const int wchunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    auto dma = [&](const T* src, unsigned lds_byte_addr) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(src), "s"(lds_byte_addr)
                   : "memory");
    };
    const T* wlane = nullptr;                      // this lane's 16 bytes of (row wave * 8 + lane / 8, tap 0, chunk 0) of the tile's rows
    int cfirst = 0, nchunks = 0, nkt = 0;               // this work item's first chunk, chunk count, K-steps
    // (a per-work-item rotation of the chunk walk — L2 channel spread of lockstep workgroups — was measured in round 5: 1.5 % SLOWER end to end, profiles/r5_t160_findings.md)
    auto issue_w = [&](int cc, int tap, int stage) {
      const T* src = wlane + ((long)tap * ctot + cc * BK);
      const unsigned dst = lds0 + W_BASE + (unsigned)stage * WST_BYTES + (unsigned)wave * 1024u;
#pragma unroll
      for (int j = 0; j < WJ; ++j) dma(src + (long)j * 32 * p.K, dst + (unsigned)j * 4096u);
    };
    // slab staging: thread -> (row tid / 8 + 32 j, slot tid % 8); key(row) = (row >> 1) & 7 = (tid >> 4) & 7 for every j, so a
    // thread stages ONE 8-channel group of the chunk: its 16 GroupNorm coefficients are loaded once per chunk
    const int schunk = (tid & 7) ^ ((tid >> 4) & 7);
    const unsigned sdst = (unsigned)tid * 16u;     // byte offset of (row tid / 8, slot tid % 8) inside a slab buffer
    int spix[SJ];                                   // input pixel of this thread's slab row j (-1: zero padding / beyond the slab)
    u32x4 sreg[SJ];                                 // the chunk being staged
    f32x4 ca0, ca1, cd0, cd1;                       // its GroupNorm coefficients a[8], d[8] (of the image of patch 0)
    f32x4 cb0, cb1, ce0, ce1;                       // NP = 2: the same for the image of patch 1 (slab rows >= WIN)
    int img = 0, img1 = 0;
    auto load_slab = [&](int cc) {                  // request chunk cc of the window (asm: the compiler's waitcnt pass must not see these)
      int c = cc * BK;
      const T* base = A0;
      int pitch = p.c0;
      if (c >= p.c0) { base = A1; pitch = p.c1; c -= p.c0; }
      c += schunk * 8;
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        const T* src = spix[j] >= 0 ? base + (long)spix[j] * pitch + c : zero;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(sreg[j]) : "v"(src) : "memory");
      }
      if constexpr (PRO) {
        const float* ca = coef + (long)img * 2 * ctot + cc * BK + schunk * 8;
        const float* cd = ca + ctot;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(ca0) : "v"(ca) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=&v"(ca1) : "v"(ca) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(cd0) : "v"(cd) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=&v"(cd1) : "v"(cd) : "memory");
        if constexpr (NP == 2) {
          const float* cb = coef + (long)img1 * 2 * ctot + cc * BK + schunk * 8;
          const float* ce = cb + ctot;
          asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(cb0) : "v"(cb) : "memory");
          asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=&v"(cb1) : "v"(cb) : "memory");
          asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(ce0) : "v"(ce) : "memory");
          asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=&v"(ce1) : "v"(ce) : "memory");
        }
      }
    };
    constexpr int NCOEF = PRO ? 4 * NP : 0;        // coefficient loads per chunk (vmcnt accounting)
    // after a `s_waitcnt vmcnt` that covers the loads above: ties every later use of the staged registers to this point
    auto slab_landed = [&]() {
#pragma unroll
      for (int j = 0; j < SJ; ++j) asm volatile("" : "+v"(sreg[j]));
      if constexpr (PRO) asm volatile("" : "+v"(ca0), "+v"(ca1), "+v"(cd0), "+v"(cd1));
      if constexpr (PRO && NP == 2) asm volatile("" : "+v"(cb0), "+v"(cb1), "+v"(ce0), "+v"(ce1));
    };
    const bool silu = p.a_silu != 0;
    auto xform_piece = [&](int j) {                 // normalise (+SiLU) slab row j in its registers (zero padding stays zero)
      if constexpr (PRO) {
        V8 v = __builtin_bit_cast(V8, sreg[j]);
        float a[8] = {ca0[0], ca0[1], ca0[2], ca0[3], ca1[0], ca1[1], ca1[2], ca1[3]};
        float d[8] = {cd0[0], cd0[1], cd0[2], cd0[3], cd1[0], cd1[1], cd1[2], cd1[3]};
        if constexpr (NP == 2) {                    // slab rows >= WIN belong to patch 1 (possibly another image)
          const bool second = (tid >> 3) + 32 * j >= WIN;
          const float b[8] = {cb0[0], cb0[1], cb0[2], cb0[3], cb1[0], cb1[1], cb1[2], cb1[3]};
          const float f[8] = {ce0[0], ce0[1], ce0[2], ce0[3], ce1[0], ce1[1], ce1[2], ce1[3]};
#pragma unroll
          for (int e = 0; e < 8; ++e) { a[e] = second ? b[e] : a[e]; d[e] = second ? f[e] : d[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = to_f32<T>(v[e]) * a[e] + d[e];   // tg_norm.hip gn_apply_kernel / gn_small_kernel: the same expression
          v[e] = from_f32<T>(silu ? silu_f(f) : f);
        }
        u32x4 r = __builtin_bit_cast(u32x4, v);
        const bool ok = spix[j] >= 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = ok ? r[e] : 0u;
        sreg[j] = r;
      }
    };
    auto write_slab = [&]() {
#pragma unroll
      for (int j = 0; j < SJ; ++j) *reinterpret_cast<u32x4*>(smem + sdst + (unsigned)j * 4096u) = sreg[j];
    };
    auto setup_tile = [&](int v) {
      int t, sp;
      work_item(v, t, sp);
      const int tile_n = t % p.tiles_n, tile_m = t / p.tiles_n;
      const long m0 = (long)tile_m * BM, n0 = (long)tile_n * BN;
      cfirst = sp * cps;
      nchunks = nchunks_all - cfirst < cps ? nchunks_all - cfirst : cps;
      nkt = nchunks * 9;
      // PATCH TILES (round 3): a tile is TH rows x WI columns of an image that may be WIDER than WI (p.in_w = 128 with WI = 64: SDXL's
      // 128 x 128 level; p.in_w = 96 with WI = 32: SD-2.1's 96 x 96 level): tile_m -> (image, patch row ty, patch column tx); the
      // window's halo columns then hold real neighbour pixels instead of padding.  p.in_w == WI is the whole-rows case of round 2.
      // generic patches (round 3, second pass): any in_w that is a multiple of WI (48 = 3 x 16, 24 = 3 x 8), NP patches per tile.
      const int tpr = p.in_w / WI, tpi = (H / TH) * tpr;       // patches per image row / per image
      const int g0 = tile_m * NP;
      img = g0 / tpi;
      const int rem = g0 - img * tpi;
      const int y0 = (rem / tpr) * TH, x0 = (rem % tpr) * WI;
      int y1 = 0, x1 = 0;
      if constexpr (NP == 2) {
        img1 = (g0 + 1) / tpi;
        const int rem1 = g0 + 1 - img1 * tpi;
        y1 = (rem1 / tpr) * TH; x1 = (rem1 % tpr) * WI;
      }
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        const int sr = (tid >> 3) + 32 * j;
        const bool second = NP == 2 && sr >= WIN;
        const int r = second ? sr - WIN : sr;
        const int sy = r / SW, sx = r - sy * SW;
        const int iy = (second ? y1 : y0) - 1 + sy, ix = (second ? x1 : x0) - 1 + sx;
        const bool ok = sr < SLAB && iy >= 0 && iy < H && ix >= 0 && ix < p.in_w;
        spix[j] = ok ? ((second ? img1 : img) * H + iy) * p.in_w + ix : -1;
      }
      wlane = Wp + (n0 + wave * 8 + (lane >> 3)) * p.K + wchunk * 8;
    };
    // tile prologue: weight tiles 0..2, window chunk 0 into registers (normalised there); for every tile but a workgroup's first
    // this runs while the compute waves are in the previous tile's epilogue (which bounces through the slab region)
    auto prologue = [&](int v) {
      setup_tile(v);
      issue_w(cfirst, 0, 0);
      load_slab(cfirst);
      issue_w(cfirst, 1, 1);
      issue_w(cfirst, 2, 2);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WJ) : "memory");  // all but weight tiles 1 and 2
      slab_landed();
#pragma unroll
      for (int j = 0; j < SJ; ++j) xform_piece(j);
    };

    int v = blockIdx.x;
    if (v < ntiles) prologue(v);
    for (; v < ntiles; v += gridDim.x) {
      __builtin_amdgcn_s_barrier();                 // S1: the compute waves are out of their epilogue, the slab region is free
      write_slab();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                 // S2: window chunk 0 and weight tile 0 are in place
      int kt = 0;
      for (int cc = 0; cc < nchunks; ++cc) {
        const bool more = cc + 1 < nchunks;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap, ++kt) {
          // the next chunk's window: requested at tap 0, landed by the seam of tap 2, normalised in registers in taps 3..8
          // (two rows in taps 3..5, one in taps 6..8: a loader never holds up a barrier with a long VALU stretch)
          if (tap == 0 && more) load_slab(cfirst + cc + 1);
          if (more) {
#pragma unroll
            for (int j = 0; j < SJ; ++j)
              if ((tap >= 3 && tap <= 5 && j / 2 == tap - 3) || (tap >= 6 && j == tap)) xform_piece(j);
          }
          // seam of K-step kt: weight tile kt + 1 (requested TWO K-steps ago) has landed; tile kt + 2 and, in taps 0 and 1, the
          // window loads (younger than tile kt + 2 in tap 0, than tile kt + 1 in neither) may stay in flight
          if (kt + 2 < nkt) {
            if (tap <= 1 && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WJ + SJ + NCOEF) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WJ) : "memory");
          } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (tap == 2 && more) slab_landed();
          __builtin_amdgcn_s_barrier();
          if (kt + 3 < nkt) {                       // every compute wave has its last fragments of stage tap % 3: refill it
            const int t3 = tap + 3;
            issue_w(cfirst + (t3 >= 9 ? cc + 1 : cc), t3 >= 9 ? t3 - 9 : t3, tap % 3);
          }
          if (tap == 8 && more) {                   // chunk boundary: the slab has been read for the last time
            write_slab();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();           // X: the next chunk's window is in place
          }
        }
      }
      const int vn = v + (int)gridDim.x;
      if (vn < ntiles) prologue(vn);                // under the compute waves' epilogue
    }
    return;
  }

  // ============================================================= compute waves =============================================
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const unsigned rkey = (unsigned)((l31 >> 1) & 7);
  const unsigned fw0 = lds0 + W_BASE + (unsigned)((wave_n * TN * 32 + l31) * 128) + (((unsigned)hi ^ rkey) << 4);
  int srow[TM];                                     // slab row of this lane's pixel for tap (0, 0)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int pm = wave_m * TM * 32 + i * 32 + l31;
    const int pk = pm / PP, pq = pm - pk * PP;      // patch of the tile, pixel inside it
    srow[i] = pk * WIN + (pq / WI) * SW + pq % WI;
  }
  unsigned ax[TM], aw;                              // k-step 0 addresses of the current K-step
  auto set_addr = [&](int tap) {                    // K-step (any chunk, tap): slab rows of the tap, weight stage tap % 3 (9 taps per chunk)
    const int off = (tap / 3) * SW + tap % 3;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const unsigned sr = (unsigned)(srow[i] + off);
      ax[i] = lds0 + sr * 128u + ((((sr >> 1) & 7u) ^ (unsigned)hi) << 4);
    }
    aw = fw0 + (unsigned)(tap % 3) * WST_BYTES;
  };
  // Fragment pipeline (inline asm reads: they stay where they are written; hand-counted lgkmcnt, LDS returns in order).  The
  // 256-register budget of two waves per SIMD holds 160 accumulators, so the weight fragments are NOT double-buffered: w[j] is
  // used by the two MFMAs of column j and re-read for the next k-step right behind them (10 MFMAs = 320 cycles before its next
  // use); the two pixel fragments alternate between two sets, read at the top of the previous k-step.  Reads issued after
  // w[j] of k-step s and before its use: w[j+1..4] of s, x of s + 1, w[0..j-1] of s + 1 = 6 for every j -> lgkmcnt(6).
  auto read_x = [&](u32x4 (&xf)[TM], int ks) {
    const unsigned kx = (unsigned)ks << 5;          // chunk 2 ks + hi: flips bits 5..6 of the swizzled slot (bases are 128-byte aligned)
    asm volatile("ds_read_b128 %0, %1" : "=v"(xf[0]) : "v"(ax[0] ^ kx));
    asm volatile("ds_read_b128 %0, %1" : "=v"(xf[1]) : "v"(ax[1] ^ kx));
  };
  auto read_w = [&](u32x4& wf, int j, int ks) {
    const unsigned a = aw ^ ((unsigned)ks << 5);
    if (j == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(wf) : "v"(a));
    if (j == 1) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(wf) : "v"(a));
    if (j == 2) asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(wf) : "v"(a));
    if (j == 3) asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(wf) : "v"(a));
    if (j == 4) asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(wf) : "v"(a));
  };
  // one k-step: x_next <- pixel fragments of k-step `nks`, then per column j: wait, 2 MFMAs, re-read w[j] for k-step `nks`
  // (addresses ax / aw must already point at the K-step that `nks` belongs to)
  auto kstep = [&](f32x16 (&acc)[TM][TN], const u32x4 (&xc)[TM], u32x4 (&xn)[TM], u32x4 (&wf)[TN], int nks, bool have_next,
                   bool with_x = true) {
    if (have_next && with_x) read_x(xn, nks);
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

  for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
    int t, sp;
    work_item(v, t, sp);
    const int lbid = t * S + sp;                    // index of this work item's fp32 partial (reduce kernel: tile-major, split-minor)
    const int tile_n = t % p.tiles_n, tile_m = t / p.tiles_n;
    const long m0 = (long)tile_m * BM, n0 = (long)tile_n * BN;
    const int nchunks = nchunks_all - sp * cps < cps ? nchunks_all - sp * cps : cps;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    __builtin_amdgcn_s_barrier();                   // S1 (see the loader)
    __builtin_amdgcn_s_barrier();                   // S2
    __builtin_amdgcn_sched_barrier(0);
    u32x4 xa[TM], xb[TM], wf[TN];
    set_addr(0);
    read_x(xa, 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) read_w(wf[j], j, 0);
    for (int cc = 0; cc < nchunks; ++cc) {
      const bool more = cc + 1 < nchunks;
      // the 18 per-tap slab addresses are invariant over the chunk loop: hoisted they cost 18 registers (and spills at the
      // 256-register budget); opaque per iteration they are 6 VALU instructions per K-step
      asm volatile("" : "+v"(srow[0]), "+v"(srow[1]));
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        kstep(acc, xa, xb, wf, 1, true);
        kstep(acc, xb, xa, wf, 2, true);
        kstep(acc, xa, xb, wf, 3, true);
        // seam: all my reads of this weight stage (and, at tap 8, of the slab) were issued above: wait for them, then the
        // workgroup barrier; behind it the loaders refill the stage and the last k-step reads the NEXT K-step's fragments
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const bool last = !more && tap == 8;
        if (!last) set_addr(tap == 8 ? 0 : tap + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (tap == 8) {
          // chunk boundary: the loaders are rewriting the slab: the next weight fragments now, the pixel fragments behind barrier X
          kstep(acc, xb, xa, wf, 0, !last, false);
          if (more) {
            __builtin_amdgcn_s_barrier();           // X
            __builtin_amdgcn_sched_barrier(0);
            read_x(xa, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          kstep(acc, xb, xa, wf, 0, true);
        }
      }
    }
    // every wave is past the last barrier with all its fragment reads done; the next tile's prologue leaves the slab region alone
    // (measured and dropped, scripts/dev_slab_exp.py same-process A/B: the loaders touching the tile's residual lines a few
    // K-steps ahead so that the epilogue's loads hit L2: 2 % slower; s_setprio on either role: no gain)
    // token index of the wave's first pixel and the distance between its two 32-row blocks.  Whole-row tiles: m0 + 64 wave_m, 32.
    // Patch tile (p.in_w > WI, never split): the wave's 64 pixels are 64 / WI patch rows of WI contiguous tokens each; the two blocks
    // are 32 tokens apart inside one patch row (WI = 64) or one image row apart (WI = 32).
    // Patch tiles (p.patch_pwl > 0): the wave's first pixel is patch_token(tile, 64 wave_m); 64-pixel patch rows: blocks 32 tokens apart;
    // 32-pixel patch rows: one image row apart; 16- / 8-pixel patch rows: per-row mapping (wave_row_token's pwl form).  A SPLIT work
    // item writes its fp32 partial in tile-local row order (the reduce kernel maps rows to tokens with the same patch_token).
    long mw = m0 + wave_m * TM * 32, mbase = m0;
    const bool patch = PATCH && S == 1;
    if (patch) {
      mw = patch_token(p, tile_m, wave_m * 64);
      mbase = mw;
    }
    float* scr = reinterpret_cast<float*>(smem) + wave * (32 * 68);
    const long nw = n0 + wave_n * TN * 32;
    const int part = S > 1 ? lbid : -1;
    // one call per row mapping, so that the whole-row tiles keep a compile-time block distance (the row loop's address arithmetic)
    if constexpr (WI == 64 || !PATCH) {
      epilogue_tile_lds<T, TM, TN, 0>(p, acc, mw, nw, lane, scr, part, mbase, n0);
    } else if constexpr (WI == 32) {
      if (patch) epilogue_tile_lds<T, TM, TN, 0>(p, acc, mw, nw, lane, scr, -1, mbase, n0, (long)p.in_w);
      else epilogue_tile_lds<T, TM, TN, 0>(p, acc, mw, nw, lane, scr, part, mbase, n0);
    } else {
      if (patch) epilogue_tile_lds<T, TM, TN, 0, false, true>(p, acc, mw, nw, lane, scr, -1, mbase, n0, 32, p.patch_pwl);
      else epilogue_tile_lds<T, TM, TN, 0>(p, acc, mw, nw, lane, scr, part, mbase, n0);
    }
  }
}

template <typename T, int WI, int NP, bool PRO, bool PATCH>
int launch_slab(const tg_gemm_desc* d, GemmParams p, int splits, hipStream_t st) {
  constexpr int BM = 128, TH = BM / NP / WI, SLAB = NP * (TH + 2) * (WI + 2), SJ = (SLAB + 31) / 32;
  constexpr size_t slab = (size_t)SJ * 32 * 128, scratch = 4 * 32 * 68 * 4;
  const size_t lds = (slab > scratch ? slab : scratch) + 3 * (size_t)320 * 128;
  const long tiles_m = d->M / BM, tiles_n = d->N / 320;
  const int nchunks = (d->c0 + (d->a1 ? d->c1 : 0)) / BK;
  p.tiles_n = (int)tiles_n;
  p.full_tiles = 0;                               // with splits > 1 every tile is a "tail" tile of the reduce kernel
  p.tail_s = splits;
  p.kt_per_split = (nchunks + splits - 1) / splits;   // channel chunks per split
  p.tile_bm = BM; p.tile_bn = 320;
  // weight-heavy layer: (column tile, split)-major work order (see work_item); TG_GEMM_FLAGS bit 13 (dev A/B) keeps the tile-major order
  p.slab_order = ((long)d->N * d->K > (long)d->M * (d->K / 9) && !(p.flags & 8192)) ? 1 : 0;
  long grid = tiles_m * tiles_n * splits;
  if (grid > 256) grid = 256;                     // one persistent workgroup per CU
  auto k = conv_slab_kernel<T, WI, NP, PRO, PATCH>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr;
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(512), lds, st, p);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

template <typename T>
int launch_slab_dtype(const tg_gemm_desc* d, const GemmParams& p, int splits, hipStream_t st) {
  // the planner (tg_gemm.hip: slab_geometry) chose the patch width / patches per tile: whole rows of 64 / 32 / 16-wide maps, patches of
  // wider maps (128 = 2 x 64, 96 = 3 x 32, 48 = 3 x 16, ...), two 8 x 8 patches for the 8- / 24- / 40-wide maps
  const bool patch = p.patch_pwl > 0;
  const int pw = patch ? (1 << p.patch_pwl) : d->out_w;
  const bool pro = d->a_coef != nullptr;
#define TG_SLAB_CASE(W, N, PT)                                                                                         \
  if (pw == W && p.patch_np == N && patch == PT)                                                                         \
    return pro ? launch_slab<T, W, N, true, PT>(d, p, splits, st) : launch_slab<T, W, N, false, PT>(d, p, splits, st);
  TG_SLAB_CASE(64, 1, false)
  TG_SLAB_CASE(32, 1, false)
  TG_SLAB_CASE(16, 1, false)
  TG_SLAB_CASE(64, 1, true)
  TG_SLAB_CASE(32, 1, true)
  TG_SLAB_CASE(16, 1, true)
  TG_SLAB_CASE(8, 2, true)
#undef TG_SLAB_CASE
  tg_set_error("tg_gemm conv: no slab kernel for width %d (patch %d x %d)", d->out_w, pw, p.patch_np);
  return TG_ERR_UNSUPPORTED;
}

}  // namespace

// ping-pong compute waves (tg_conv_slab_pp.hip, round 6): whole-row tiles of the 64 / 32 / 16-wide maps
int tg_conv_slab_pp_launch(const tg_gemm_desc* d, const void* params, int splits, void* stream);

// Called by tg_gemm.hip's planner (not part of the C ABI); GemmParams arrives filled except for the tile bookkeeping.  With
// splits > 1 the caller runs the reduce kernel over the tiles_m * tiles_n tail tiles of `splits` partials each.
// TG_SLAB_PP (dev A/B knob): 0 keeps every layer on conv_slab_kernel (one compute wave per SIMD), 1 = the two-waves-per-SIMD kernel on the 64-wide
// maps only, 2 = on the 32- and 16-wide maps as well, 3 (default) = also on the 64 / 32 / 16-wide PATCH tiles of wider maps.  Isolated launches (scripts/dev_slab_pp.py): 64 x 64 layers 4-6 % faster, 32 x 32 equal,
// 16 x 16 (split tiles) 5-10 % slower; under graph replay, same box, interleaved (profiles/r6_ab_slab2w.json): 864.3 -> 859.3 (mode 1) -> 855.3 ms per
// story (mode 2, +1.05 %): the whole-step evidence decides.
bool tg_conv_slab_is_pp(const tg_gemm_desc* d, int patch_pwl, int patch_np, int epi_lds) {
  const char* e = getenv("TG_SLAB_PP");
  const int mode = e == nullptr ? 3 : (int)strtol(e, nullptr, 0);
  // tile width: the whole row, or (round 6, TG_SLAB_PP >= 3 = default) one patch of a wider map (SD-2.1's 96 / 48-wide, SDXL's 128-wide levels)
  const int pw = patch_pwl > 0 ? (1 << patch_pwl) : d->out_w;
  if (patch_pwl > 0 && mode < 3) return false;
  const bool w_ok = pw == 64 || (mode >= 2 && (pw == 32 || pw == 16));
  return mode >= 1 && w_ok && patch_np == 1 && epi_lds && d->in_w == d->out_w;
}

int tg_conv_slab_launch(const tg_gemm_desc* d, const void* params, int splits, void* stream) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  if (tg_conv_slab_is_pp(d, p.patch_pwl, p.patch_np, p.epi_lds)) return tg_conv_slab_pp_launch(d, params, splits, stream);
  TG_CHECK(p.gn_part == nullptr, TG_ERR_UNSUPPORTED, "tg_gemm conv: out_gn_partials needs the two-wave slab kernel (tg_gemm_gn_partial_blocks)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == TG_BF16) return launch_slab_dtype<bf16_t>(d, p, splits, st);
  return launch_slab_dtype<f16_t>(d, p, splits, st);
}

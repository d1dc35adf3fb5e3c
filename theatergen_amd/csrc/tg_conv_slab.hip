// 3x3 convolution for the 320-multiple channel counts of the SD UNets: one output tile = BM pixels x 320 output channels per
// workgroup, the input window staged ONCE per 320 channels, GroupNorm(+SiLU) applied while it is staged
// (include/theatergen_hip.h: tg_gemm mode 1, selected by the planner in tg_gemm.hip; replaces the conv1 / conv2 +
// nonlinearity(norm(x)) pairs of ResnetBlock2D, models/resnet.py via models/unet_2d_blocks.py:184-195).
//
// Why a second conv kernel.  conv_halo_kernel (tg_gemm.hip) gives every 128-channel output tile its own workgroup: N = 320
// costs three tiles (the third half empty: 17 % of the MFMA work wasted), every tile re-stages the input window (measured
// 272 MB per launch against 86 MB algorithmic on the 64^2 320 -> 320 layer), the window of the next channel chunk can only be
// requested after the last read of the current one (single slab buffer), and because the slab goes HBM -> LDS by DMA nothing
// can be applied to it on the way: GroupNorm + SiLU was a separate pass that wrote the normalised tensor to HBM and read it
// back.  Here:
//   * 4 waves, ONE wave per SIMD with the whole 512-register file: wave tile (BM / 2) x 160 = TM x 5 MFMA tiles of 32 x 32
//     (160 accumulator registers at BM = 128); 7 fragment reads feed 10 MFMAs per k-step, LDS read time is 1/3 of MFMA time;
//   * the input window (slab: (BM / W + 2) x (W + 2) pixels x 64 channels, 128-byte swizzled rows) is double-buffered and goes
//     global -> registers -> LDS: requested at tap 8 two chunks ahead, normalised (x * a[b, c] + d[b, c], SiLU — the same fp32
//     expression and rounding point as tg_groupnorm's apply pass, so the bf16 / fp16 MFMA inputs are bit-identical to the unfused
//     path) one element per second MFMA of taps 1..5 (the VALU work rides in the matrix pipe's shadow instead of in a block of
//     its own) and written there, first read at tap 0 of the next chunk;
//   * weight tiles (320 x 64, 40 KB) by LDS-DMA into two stages; a K-step is 4 k-steps x 10 MFMAs = 1280 matrix-pipe cycles, ONE
//     barrier per K-step placed before the last k-step (as in tg_gemm_bt.hip), fragments double-buffered in registers by
//     inline-asm ds_read_b128 with hand-counted lgkmcnt;
//   * persistent workgroups, one per CU, XCD-chunked tile order; epilogue = the shared LDS-transposed one (bias + time-embedding
//     vector + residual + scale, 16-byte stores on whole 128-byte rows).
// K order (channel chunk, tap, k) is conv_halo_kernel's: the two kernels produce bit-identical results on the same input.
#include "tg_gemm_common.h"

namespace {

template <typename T, int WI, bool PRO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_slab_kernel(GemmParams p) {
  constexpr int BM = 128, BN = 320, TM = BM / 64, TN = 5, NF = TM + TN;
  constexpr int TH = BM / WI, SW = WI + 2, SROWS = TH + 2, SLAB = SROWS * SW, SJ = (SLAB + 31) / 32;
  constexpr unsigned SLAB_BYTES = SJ * 32 * 128, WST_BYTES = BN * 128, W_BASE = 2 * SLAB_BYTES;
  constexpr int WJ = BN / 32;                      // LDS-DMA instructions per wave per weight tile (8 rows x 128 B each)
  static_assert(BM % WI == 0 && TM == 2, "whole image rows per tile");
  static_assert(SJ <= 9, "two slab rows per thread in taps 1 and 2, one in taps 3..7");
  static_assert(4 * 32 * 68 * 4 <= 2 * WST_BYTES, "epilogue bounce fits the weight stages");
  typedef typename Vec<T>::v8 V8;

  extern __shared__ __attribute__((aligned(128))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;

  const T* A0 = reinterpret_cast<const T*>(p.a0);
  const T* A1 = reinterpret_cast<const T*>(p.a1);
  const T* Wp = reinterpret_cast<const T*>(p.w);
  const T* zero = reinterpret_cast<const T*>(tg_zero_page);
  const float* coef = reinterpret_cast<const float*>(p.a_coef);
  const int ctot = p.c0 + p.c1;
  const int nchunks = ctot / BK;
  const int nkt = nchunks * 9;
  const int H = p.in_h;
  const int tiles_m = (int)(p.M / BM);
  const int ntiles = tiles_m * p.tiles_n;

  // ---- weight tiles: LDS-DMA, instruction q = j * 4 + wave covers rows [8q, 8q + 8): lane -> (row 8q + lane / 8, slot lane % 8),
  // chunk fetched into a slot = slot ^ key(row), key(row) = (row >> 1) & 7 (conflict-free ds_read_b128 of 32 consecutive rows)
  const int wchunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  auto dma = [&](const T* src, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_byte_addr)
                 : "memory");
  };
  const T* wlane = nullptr;                        // this lane's 16 bytes of (row wave * 8 + lane / 8, tap 0, chunk 0) of the tile's rows
  auto issue_w = [&](int cc, int tap, int stage) {
    const T* src = wlane + ((long)tap * ctot + cc * BK);
    const unsigned dst = lds0 + W_BASE + (unsigned)stage * WST_BYTES + (unsigned)wave * 1024u;
#pragma unroll
    for (int j = 0; j < WJ; ++j) dma(src + (long)j * 32 * p.K, dst + (unsigned)j * 4096u);
  };

  // ---- slab staging: thread -> (row tid / 8 + 32 j, slot tid % 8); key(row) = (row >> 1) & 7 = (tid >> 4) & 7 for every j, so
  // a thread stages ONE 8-channel group of the chunk: its 16 GroupNorm coefficients are loaded once per chunk
  const int schunk = (tid & 7) ^ ((tid >> 4) & 7);
  const unsigned sdst = (unsigned)tid * 16u;       // byte offset of (row tid / 8, slot tid % 8) inside a slab buffer
  int spix[SJ];                                     // input pixel of this thread's slab row j (-1: zero padding / beyond the slab)
  u32x4 sreg[SJ];                                   // the chunk being staged
  f32x4 ca0, ca1, cd0, cd1;                         // its GroupNorm coefficients a[8], d[8]
  int img = 0;
  auto load_slab = [&](int cc) {                    // request chunk cc of the window (asm: the compiler's waitcnt pass must not see these)
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
    }
  };
  // after a `s_waitcnt vmcnt` that covers the loads above: ties every later use of the staged registers to this point
  auto slab_landed = [&]() {
#pragma unroll
    for (int j = 0; j < SJ; ++j) asm volatile("" : "+v"(sreg[j]));
    if constexpr (PRO) asm volatile("" : "+v"(ca0), "+v"(ca1), "+v"(cd0), "+v"(cd1));
  };
  const bool silu = p.a_silu != 0;
  auto xform = [&](const u32x4& r, int e) -> float {   // tg_norm.hip gn_apply_kernel / gn_small_kernel: the same fp32 expression
    const V8 v = __builtin_bit_cast(V8, r);
    const float a = e < 4 ? ca0[e & 3] : ca1[e & 3], d = e < 4 ? cd0[e & 3] : cd1[e & 3];
    const float f = to_f32<T>(v[e]) * a + d;
    return silu ? silu_f(f) : f;
  };
  auto store_piece = [&](int j, int buf, const float (&fe)[8]) {    // slab row tid / 8 + 32 j (zero padding stays zero)
    V8 v = __builtin_bit_cast(V8, sreg[j]);
    if constexpr (PRO) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = from_f32<T>(fe[e]);
      u32x4 r = __builtin_bit_cast(u32x4, v);
      const bool ok = spix[j] >= 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = ok ? r[e] : 0u;
      v = __builtin_bit_cast(V8, r);
    }
    *reinterpret_cast<V8*>(smem + (unsigned)buf * SLAB_BYTES + sdst + (unsigned)j * 4096u) = v;
  };
  auto stage_piece = [&](int j, int buf) {          // all at once (tile prologue)
    float fe[8];
    if constexpr (PRO) {
#pragma unroll
      for (int e = 0; e < 8; ++e) fe[e] = xform(sreg[j], e);
    }
    store_piece(j, buf, fe);
  };
  float fe[8];                                      // the slab row being normalised under the MFMAs
  // filler after MFMA number s (0..19) of a k-step pair: element s / 2 of slab row j after every second MFMA, the store after the 17th
  auto fill = [&](int s, int j, int buf) {
    if (j < 0 || j >= SJ) return;
    if constexpr (PRO) {
      if (s < 16 && (s & 1) == 0) fe[s >> 1] = xform(sreg[j], s >> 1);
    }
    if (s == 16) store_piece(j, buf, fe);
  };

  // ---- fragment reads (inline asm: they stay where they are written; counted lgkmcnt waits below)
  const unsigned rkey = (unsigned)((l31 >> 1) & 7);
  const unsigned fw0 = lds0 + W_BASE + (unsigned)((wave_n * TN * 32 + l31) * 128) + (((unsigned)hi ^ rkey) << 4);
  int srow[TM];                                     // slab row of this lane's pixel for tap (0, 0)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int pm = wave_m * TM * 32 + i * 32 + l31;
    srow[i] = (pm / WI) * SW + pm % WI;
  }
  unsigned ax[TM], aw;                              // k-step 0 addresses of the current K-step
  auto set_addr = [&](int tap, int buf, int stage) {
    const int off = (tap / 3) * SW + tap % 3;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const unsigned sr = (unsigned)(srow[i] + off);
      ax[i] = lds0 + (unsigned)buf * SLAB_BYTES + sr * 128u + ((((sr >> 1) & 7u) ^ (unsigned)hi) << 4);
    }
    aw = fw0 + (unsigned)stage * WST_BYTES;
  };
  auto read_frags = [&](u32x4 (&xf)[TM], u32x4 (&wf)[TN], int ks) {
    const unsigned kx = (unsigned)ks << 5;          // chunk 2 ks + hi: flips bits 5..6 of the swizzled slot (bases are 128-byte aligned)
    asm volatile("ds_read_b128 %0, %1" : "=v"(xf[0]) : "v"(ax[0] ^ kx));
    if constexpr (TM >= 2) asm volatile("ds_read_b128 %0, %1" : "=v"(xf[TM >= 2 ? 1 : 0]) : "v"(ax[TM >= 2 ? 1 : 0] ^ kx));
    const unsigned a = aw ^ kx;
    asm volatile("ds_read_b128 %0, %1" : "=v"(wf[0]) : "v"(a));
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(wf[1]) : "v"(a));
    asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(wf[2]) : "v"(a));
    asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(wf[3]) : "v"(a));
    asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(wf[4]) : "v"(a));
  };
  auto mfmas = [&](f32x16 (&acc)[TM][TN], const u32x4 (&xf)[TM], const u32x4 (&wf)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(__builtin_bit_cast(V8, wf[j]), __builtin_bit_cast(V8, xf[i]), acc[i][j]);
  };
  auto mfmas_fill = [&](f32x16 (&acc)[TM][TN], const u32x4 (&xf)[TM], const u32x4 (&wf)[TN], int s0, int j, int buf) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) {
        acc[i][jn] = mfma32(__builtin_bit_cast(V8, wf[jn]), __builtin_bit_cast(V8, xf[i]), acc[i][jn]);
        fill(s0 + i * TN + jn, j, buf);
        __builtin_amdgcn_sched_barrier(0);
      }
  };
#define CS_LGKM(N)                                             \
  do {                                                         \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); \
    __builtin_amdgcn_sched_barrier(0);                         \
  } while (0)

  for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
    const int lbid = xcd_chunked_block_id(v, ntiles);
    const int tile_n = lbid % p.tiles_n, tile_m = lbid / p.tiles_n;
    const long m0 = (long)tile_m * BM, n0 = (long)tile_n * BN;
    img = (int)(m0 / ((long)H * WI));
    const int y0 = (int)((m0 - (long)img * H * WI) / WI);
#pragma unroll
    for (int j = 0; j < SJ; ++j) {
      const int sr = (tid >> 3) + 32 * j;
      const int sy = sr / SW, sx = sr - sy * SW;
      const int iy = y0 - 1 + sy, ix = sx - 1;
      const bool ok = sr < SLAB && iy >= 0 && iy < H && ix >= 0 && ix < WI;
      spix[j] = ok ? (img * H + iy) * WI + ix : -1;
    }
    wlane = Wp + (n0 + wave * 8 + (lane >> 3)) * p.K + wchunk * 8;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- tile prologue: weight tile 0, window chunk 0 (through registers), weight tile 1, window chunk 1 (stays in registers)
    issue_w(0, 0, 0);
    load_slab(0);
    issue_w(0, 1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WJ) : "memory");        // all but weight tile 1
    slab_landed();
#pragma unroll
    for (int j = 0; j < SJ; ++j) stage_piece(j, 0);
    if (nchunks > 1) load_slab(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    u32x4 xa[TM], wa[TN], xb[TM], wb[TN];
    set_addr(0, 0, 0);
    read_frags(xa, wa, 0);
    int kt = 0;
    for (int cc = 0; cc < nchunks; ++cc) {
      const int buf = cc & 1;
      const bool more = cc + 1 < nchunks;
      const bool ahead = cc + 2 < nchunks;          // request the window of chunk cc + 2 at tap 8 (the staging registers are free then)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap, ++kt) {
        // slab rows of chunk cc + 1 normalised and written under this K-step's MFMAs: two in taps 1 and 2, one in taps 3..7
        // (on the last chunk the other buffer is dead: what is written there is never read)
        const int ja = tap == 1 ? 0 : tap == 2 ? 2 : (tap >= 3 && tap <= 7) ? tap + 1 : -1;
        const int jb = tap == 1 ? 1 : tap == 2 ? 3 : -1;
        if (tap == 8 && ahead) load_slab(cc + 2);
        read_frags(xb, wb, 1);                      // k-step 0 on fragments a; the reads of k-step 1 are in flight under it
        CS_LGKM(NF);
        mfmas_fill(acc, xa, wa, 0, ja, buf ^ 1);
        read_frags(xa, wa, 2);
        CS_LGKM(NF);
        mfmas_fill(acc, xb, wb, 10, ja, buf ^ 1);
        read_frags(xb, wb, 3);
        CS_LGKM(NF);
        mfmas_fill(acc, xa, wa, 0, jb, buf ^ 1);
        // seam: my reads of this weight stage (and, at tap 8, of this slab buffer) are complete, the next weight tile — requested
        // one K-step ago — has landed; window loads requested in THIS K-step (tap 8) stay in flight until the next seam
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const bool last = !more && tap == 8;
        if (!last) {
          if (tap == 8 && ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SJ + (PRO ? 4 : 0)) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (tap == 0) slab_landed();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (!last) {
          const int ntap = tap == 8 ? 0 : tap + 1;
          set_addr(ntap, tap == 8 ? buf ^ 1 : buf, (kt + 1) & 1);
          read_frags(xa, wa, 0);                    // first fragments of the next K-step under the last MFMAs
          if (kt + 2 < nkt) {
            const int t2 = tap + 2;                 // refill the stage just read with the tile after next
            issue_w(t2 >= 9 ? cc + 1 : cc, t2 >= 9 ? t2 - 9 : t2, kt & 1);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        mfmas_fill(acc, xb, wb, 10, jb, buf ^ 1);
      }
    }

    // every wave is past the last barrier with all its fragment reads done: the weight stages are free for the bounce
    epilogue_tile_lds<T, TM, TN, 0>(p, acc, m0 + wave_m * TM * 32, n0 + wave_n * TN * 32, lane,
                                   reinterpret_cast<float*>(smem + W_BASE) + wave * (32 * 68), -1, m0, n0);
    __builtin_amdgcn_s_barrier();                   // the next tile's weight DMA lands where the slower waves still bounce
  }
#undef CS_LGKM
}

template <typename T, int WI, bool PRO>
int launch_slab(const tg_gemm_desc* d, GemmParams p, hipStream_t st) {
  constexpr int BM = 128, TH = BM / WI, SLAB = (TH + 2) * (WI + 2), SJ = (SLAB + 31) / 32;
  const size_t lds = 2 * (size_t)SJ * 32 * 128 + 2 * (size_t)320 * 128;
  const long tiles_m = d->M / BM, tiles_n = d->N / 320;
  p.tiles_n = (int)tiles_n;
  p.full_tiles = (int)(tiles_m * tiles_n);
  p.tail_s = 1;
  p.tile_bm = BM; p.tile_bn = 320;
  long grid = tiles_m * tiles_n;
  if (grid > 256) grid = 256;                     // one persistent workgroup per CU
  auto k = conv_slab_kernel<T, WI, PRO>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr;
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, st, p);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

template <typename T>
int launch_slab_dtype(const tg_gemm_desc* d, const GemmParams& p, hipStream_t st) {
  const int w = d->out_w;
  const bool pro = d->a_coef != nullptr;
  if (w == 64) return pro ? launch_slab<T, 64, true>(d, p, st) : launch_slab<T, 64, false>(d, p, st);
  if (w == 32) return pro ? launch_slab<T, 32, true>(d, p, st) : launch_slab<T, 32, false>(d, p, st);
  tg_set_error("tg_gemm conv: no slab kernel for width %d", w);
  return TG_ERR_UNSUPPORTED;
}

}  // namespace

// Called by tg_gemm.hip's planner (not part of the C ABI); GemmParams arrives filled except for the tile bookkeeping.
int tg_conv_slab_launch(const tg_gemm_desc* d, const void* params, int bm, void* stream) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (bm != 128) { tg_set_error("tg_gemm conv: slab tiles are 128 pixels"); return TG_ERR_ARG; }
  if (d->dtype == TG_BF16) return launch_slab_dtype<bf16_t>(d, p, st);
  return launch_slab_dtype<f16_t>(d, p, st);
}

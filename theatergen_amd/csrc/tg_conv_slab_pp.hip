// Slab 3x3 convolution with TWO compute waves per SIMD (round 6; called from tg_conv_slab.hip for the whole-row tiles of the 64 / 32 / 16-wide maps).
//
// conv_slab_kernel (tg_conv_slab.hip) keeps ONE compute wave per SIMD (wave tile 64 x 160, 160 accumulator registers, 256-register budget, 47-195 spilled
// VGPRs): every fragment-read wait, barrier and weight-tile wait of that wave is matrix-pipe idle time (duty 0.39 - 0.47, profiles/r5_pmc_sq.json; ~1.0 PFLOP/s
// where the pipe alone sustains 1.88 on random operands, profiles/r6_mfma_sustained.json).  This kernel keeps that kernel's data path and barrier protocol —
// 128-pixel x 320-channel tiles, the input window staged ONCE per 64-channel chunk through registers with GroupNorm (+ SiLU) applied on the way, weight tiles
// by LDS-DMA into a ring of three 40 KB stages, four LOADER waves that never issue an MFMA, ONE workgroup barrier per K-step, persistent XCD-chunked tile walk,
// K splits as fp32 partial tiles — and gives every SIMD TWO compute waves on 16 x 16 x 32 MFMAs:
//   * wave tile 64 pixels x 80 channels = 4 x 5 MFMA tiles (80 accumulator registers; 12 waves per workgroup = three per SIMD, 168-register budget, no spills);
//   * a K-step (one tap of one 64-channel chunk) is two phases = its two 32-deep k-steps of 20 MFMAs; fragments are software-pipelined ONE PHASE ahead
//     (inline-asm ds_read_b128, hand-counted lgkmcnt): the four pixel fragments of the next phase are requested at the top of a phase (two register sets),
//     weight fragment j right behind the four MFMAs of column j; the K-step's barrier sits between its two phases (every read of the stage has returned, the
//     next tile has landed), so the second phase already reads the next K-step's first fragments;
//   * the two waves of a SIMD are synchronised by that barrier only: one fills the other's waits;
//   * epilogue: fp32 bounce through the (dead) slab region, 16 pixels x 64 channels per wave and pass (+ one 32 x 16 pass for the last 16 channels):
//     bias / time-embedding vector / residual loads and the stores are 16 bytes per lane on whole 128-byte rows; split work items store fp32 partial tiles
//     in tile-local order for tg_gemm.hip's fixed-order reduce.
// (First build of the round: the two compute groups one barrier apart, tg_gemm_pp.hip's schedule — four barriers per K-step, a phase's nine reads took ~480
// cycles against 320 of MFMA: no faster than conv_slab_kernel, scripts/dev_slab_pp.py ablations in profiles/r6_slab_findings.md.)
// K order (chunk, tap, k) and the fp32 epilogue arithmetic are conv_slab_kernel's; measured outputs are bit-identical to it (the 16 x 16 x 32 MFMA evidently
// folds a K-step's products in the same order as two 32 x 32 x 16 ones); the tests do not rely on that: they state a tolerance.
#include "tg_gemm_common.h"
#include <type_traits>

namespace {

// acc + lo * lo2 + hi * hi2 of two packed storage-dtype pairs (v_dot2c_f32_bf16 / v_dot2c_f32_f16: exact products, fp32 sum)
template <typename T> __device__ __forceinline__ float dot2acc(unsigned a, unsigned b, float acc);
template <> __device__ __forceinline__ float dot2acc<bf16_t>(unsigned a, unsigned b, float acc) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a), __builtin_bit_cast(bf2, b), acc, false);
}
template <> __device__ __forceinline__ float dot2acc<f16_t>(unsigned a, unsigned b, float acc) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b), acc, false);
}

// PATCH (own instantiations, round 6): the tile is one WI-wide patch of a wider map (image width p.in_w a multiple of WI: SD-2.1's 96 = 3 x 32 and 48 = 3 x 16, SDXL's
// 128 = 2 x 64; patches numbered image-major / patch-row / patch-column like tg_gemm_common.h: patch_token) — the window's left / right halo columns are then real
// neighbours and the epilogue's rows are not contiguous tokens.  The whole-row instances carry none of it (their register allocation is the measured one).
template <typename T, int WI, bool PRO, bool PATCH>
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) void conv_slab_pp_kernel(GemmParams p) {
  constexpr int BM = 128, BN = 320;
  constexpr int TH = BM / WI, SW = WI + 2, SROWS = TH + 2, SLAB = SROWS * SW, SJ = (SLAB + 31) / 32;
  constexpr unsigned SLAB_BYTES = SJ * 32 * 128, WST_BYTES = BN * 128, SCRATCH_BYTES = 8 * 16 * 68 * 4;
  constexpr unsigned SLAB_PAD = SLAB_BYTES > SCRATCH_BYTES ? SLAB_BYTES : SCRATCH_BYTES, W_BASE = SLAB_PAD;
  constexpr int WJ = BN / 32;                      // LDS-DMA instructions per loader wave per weight tile (8 rows x 128 B each)
  static_assert(SJ <= 9, "two slab rows per thread in taps 3..5, one in taps 6..8");
  typedef typename Vec<T>::v8 V8;

  extern __shared__ __attribute__((aligned(128))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int lane = threadIdx.x & 63;
  const int wave12 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const bool loader = wave12 >= 8;

  const int ctot = p.c0 + p.c1;
  const int nchunks_all = ctot / BK;
  const int S = p.tail_s, cps = p.kt_per_split;
  const int H = p.in_h;
  const int tiles_m = (int)(p.M / BM);
  const int ntiles = tiles_m * p.tiles_n * S;

  auto work_item = [&](int v, int& t, int& sp) {   // tg_conv_slab.hip: tile-major, or (column tile, split)-major for weight-heavy layers
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
#define SLAB_BAR()                                         \
  do {                                                     \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();                          \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)

  // dev timing switches (TG_GEMM_FLAGS, WRONG results by design; scripts/dev_slab_pp.py -> profiles/r6_slab_findings.md): 1 << 16 no weight DMA after the
  // prologue, 1 << 17 no window staging after the prologue, 1 << 18 no MFMAs
  // — compiled in only with -DTG_SLAB_DEV_BUILD (the tables of profiles/r6_slab_findings.md came from such a build); constants in the library's build
#ifdef TG_SLAB_DEV_BUILD
  const bool ab_now = (p.flags & (1 << 16)) != 0, ab_nos = (p.flags & (1 << 17)) != 0, ab_nom = (p.flags & (1 << 18)) != 0;
#else
  constexpr bool ab_now = false, ab_nos = false, ab_nom = false;
#endif
  if (loader) {
    // =========================================================== loader waves ===========================================
    const int wave = wave12 - 8;
    const int tid = (int)threadIdx.x - 512;
    const T* A0 = reinterpret_cast<const T*>(p.a0);
    const T* A1 = reinterpret_cast<const T*>(p.a1);
    const T* Wp = reinterpret_cast<const T*>(p.w);
    const T* zero = reinterpret_cast<const T*>(tg_zero_page);
    const float* coef = reinterpret_cast<const float*>(p.a_coef);
    const int wchunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    // weight requests: wave-uniform base (weight pointer + the tile's first row + the K-step's column offset + 32 rows per request: scalar adds) + this lane's
    // constant 32-bit byte offset (its row of the 8-row piece and its swizzled chunk) — no 64-bit VALU pointer arithmetic in the K loop
    auto dma_s = [&](const T* base, unsigned voff, unsigned lds_byte_addr) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_byte_addr) : "memory");
    };
    const unsigned wvoff = (unsigned)((((long)wave * 8 + (lane >> 3)) * p.K + wchunk * 8) * (long)sizeof(T));
    const T* wtile = nullptr;                         // Wp + n0 * K (wave-uniform, per tile)
    int cfirst = 0, nchunks = 0, nkt = 0;
    auto issue_w = [&](int cc, int tap, int stage) {
      const T* src = wtile + ((long)tap * ctot + cc * BK);
      const unsigned dst = lds0 + W_BASE + (unsigned)stage * WST_BYTES + (unsigned)wave * 1024u;
#pragma unroll
      for (int j = 0; j < WJ; ++j) dma_s(src + (long)j * 32 * p.K, wvoff, dst + (unsigned)j * 4096u);
    };
    const int schunk = (tid & 7) ^ ((tid >> 4) & 7);
    const unsigned sdst = (unsigned)tid * 16u;
    int spix[SJ];
    u32x4 sreg[SJ];
    f32x4 ca0, ca1, cd0, cd1;
    int img = 0;
    auto load_slab = [&](int cc) {
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
    constexpr int NCOEF = PRO ? 4 : 0;
    auto slab_landed = [&]() {
#pragma unroll
      for (int j = 0; j < SJ; ++j) asm volatile("" : "+v"(sreg[j]));
      if constexpr (PRO) asm volatile("" : "+v"(ca0), "+v"(ca1), "+v"(cd0), "+v"(cd1));
    };
    const bool silu = p.a_silu != 0;
    auto xform_piece = [&](int j) {                 // normalise (+ SiLU) slab row j in its registers: tg_norm.hip's expression and rounding point
      if constexpr (PRO) {
        V8 v = __builtin_bit_cast(V8, sreg[j]);
        const float a[8] = {ca0[0], ca0[1], ca0[2], ca0[3], ca1[0], ca1[1], ca1[2], ca1[3]};
        const float d[8] = {cd0[0], cd0[1], cd0[2], cd0[3], cd1[0], cd1[1], cd1[2], cd1[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = to_f32<T>(v[e]) * a[e] + d[e];
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
      const long n0 = (long)tile_n * BN;
      cfirst = sp * cps;
      nchunks = nchunks_all - cfirst < cps ? nchunks_all - cfirst : cps;
      nkt = nchunks * 9;
      if constexpr (PATCH) {
        const int IW = p.in_w, ppr = IW / WI, tpi = (H / TH) * ppr;
        img = tile_m / tpi;
        const int rem = tile_m - img * tpi, prow = rem / ppr;
        const int y0 = prow * TH, x0 = (rem - prow * ppr) * WI;
#pragma unroll
        for (int j = 0; j < SJ; ++j) {
          const int sr = (tid >> 3) + 32 * j;
          const int sy = sr / SW, sx = sr - sy * SW;
          const int iy = y0 - 1 + sy, ix = x0 + sx - 1;
          const bool ok = sr < SLAB && iy >= 0 && iy < H && ix >= 0 && ix < IW;
          spix[j] = ok ? (img * H + iy) * IW + ix : -1;
        }
      } else {
      const int tpi = H / TH;                       // whole-row tiles: TH image rows per tile
      img = tile_m / tpi;
      const int y0 = (tile_m - img * tpi) * TH;
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        const int sr = (tid >> 3) + 32 * j;
        const int sy = sr / SW, sx = sr - sy * SW;
        const int iy = y0 - 1 + sy, ix = sx - 1;
        const bool ok = sr < SLAB && iy >= 0 && iy < H && ix >= 0 && ix < WI;
        spix[j] = ok ? (img * H + iy) * WI + ix : -1;
      }
      }
      wtile = Wp + n0 * p.K;
    };
    auto prologue = [&](int v) {
      setup_tile(v);
      issue_w(cfirst, 0, 0);
      load_slab(cfirst);
      issue_w(cfirst, 1, 1);
      issue_w(cfirst, 2, 2);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WJ) : "memory");
      slab_landed();
#pragma unroll
      for (int j = 0; j < SJ; ++j) xform_piece(j);
    };

    int v = blockIdx.x;
    if (v < ntiles) prologue(v);
    for (; v < ntiles; v += gridDim.x) {
      SLAB_BAR();                                   // S1: the compute waves are out of their epilogue, the slab region is free
      write_slab();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SLAB_BAR();                                   // S2: window chunk 0 and weight tile 0 are in place
      int kt = 0;
      for (int cc = 0; cc < nchunks; ++cc) {
        const bool more = cc + 1 < nchunks;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap, ++kt) {
          // the next chunk's window: requested at tap 0, landed by the seam of tap 2, normalised in registers in taps 3..8
          if (tap == 0 && more && !ab_nos) load_slab(cfirst + cc + 1);
          if (more && !ab_nos) {
#pragma unroll
            for (int j = 0; j < SJ; ++j)
              if ((tap >= 3 && tap <= 5 && j / 2 == tap - 3) || (tap >= 6 && j == tap)) xform_piece(j);
          }
          // seam of K-step kt: weight tile kt + 1 (requested TWO K-steps ago) has landed; tile kt + 2 and, in taps 0 and 1, the window loads may stay in flight
          if (ab_now || ab_nos) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else if (kt + 2 < nkt) {
            if (tap <= 1 && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WJ + SJ + NCOEF) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WJ) : "memory");
          } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (tap == 2 && more) slab_landed();
          SLAB_BAR();
          if (kt + 3 < nkt && !ab_now) {            // every compute wave's reads of stage tap % 3 have returned: refill it
            const int t3 = tap + 3;
            issue_w(cfirst + (t3 >= 9 ? cc + 1 : cc), t3 >= 9 ? t3 - 9 : t3, tap % 3);
          }
          if (tap == 8 && more) {                   // chunk boundary: the window has been read for the last time
            if (!ab_nos) write_slab();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SLAB_BAR();                             // X: the next chunk's window is in place
          }
        }
      }
      const int vn = v + (int)gridDim.x;
      if (vn < ntiles) prologue(vn);                // under the compute waves' epilogue
    }
    return;
  }

  // ============================================================= compute waves =============================================
  const int group = wave12 >> 2;                   // 0: pixels 0-63 of the tile, 1: pixels 64-127; waves w and w + 4 share a SIMD
  const int wave_n = wave12 & 3;                    // 80-channel column of the tile
  const int frow = lane & 15, fq = lane >> 4;
  const unsigned fkey = (unsigned)((frow >> 1) & 7);
  unsigned aw0 = lds0 + W_BASE + (unsigned)((wave_n * 80 + frow) * 128) + ((((unsigned)fq) ^ fkey) << 4);
  // slab row of pixel tile i's row frow: (pm / WI) * SW + pm % WI with pm = group * 64 + i * 16 + frow  =  srow0 + a compile-time constant per i
  // PIXEL PERMUTATION inside a 16-pixel MFMA tile: a ds_read_b128 is served in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32): MFMA rows 0-3 / 12-15 of one
  // k-quarter share an LDS cycle with rows 4-11 of the NEXT quarter.  With the rows' 16-byte slot = (row & 1) * 8 + (piece ^ (row >> 1) & 7), slab rows u and u + 2 of one
  // aligned block of four land on the same banks when their pieces differ by one — with pixels in order that happened whenever the tile's first slab row was not a multiple
  // of four: 6 of the 9 taps (SQ_LDS_BANK_CONFLICT 22 % of the LDS-active cycles, profiles/r6_pmc_sq.json).  MFMA rows {0-3, 12-15} take the EVEN pixels of the tile and rows
  // 4-11 the ODD ones: rows two apart are then always in the same lane group and every tap reads conflict-free.  The epilogue un-permutes (scr row = fpix).
  const int fpix = (frow >= 4 && frow < 12) ? 2 * (frow - 4) + 1 : (frow < 4 ? 2 * frow : 2 * (frow - 8));
  int srow0 = ((group * 64 + fpix) / WI) * SW + (group * 64 + fpix) % WI;
  constexpr int srow_d[4] = {0, (16 / WI) * SW + 16 % WI, (32 / WI) * SW + 32 % WI, (48 / WI) * SW + 48 % WI};
  // Fragment pipeline (inline asm reads stay where they are written; LDS returns in order): see `phase` below for the issue order and the counted waits.
  // Window addresses: pixel tiles whose slab rows are 16 apart share the swizzle key (it has period 16 rows) and differ by 2048 bytes — an immediate of the read.
  // 64-wide maps: ONE address per tap (tiles at +0 / 16 / 32 / 48 rows); 32-wide: two (rows +0 / 16 and +34 / 50); 16-wide: four.  Every address instruction saved is
  // matrix-pipe time (profiles/r6_mfma_exp_overlap.json: a VALU instruction takes 1-2 cycles of it along).
  constexpr int NAX = WI == 64 ? 1 : (WI == 32 ? 2 : 4);
  unsigned ax[NAX], aw;
  auto set_x = [&](int tap) {
    const int off = (tap / 3) * SW + tap % 3;
#pragma unroll
    for (int b = 0; b < NAX; ++b) {
      const unsigned sr = (unsigned)(srow0 + srow_d[b * (4 / NAX)] + off);
      ax[b] = lds0 + sr * 128u + ((((sr >> 1) & 7u) ^ (unsigned)fq) << 4);
    }
  };
  auto set_w = [&](int tap) {
    if constexpr (WI == 16) asm volatile("" : "+v"(aw0));   // 16-wide instance: the weight-stage base is laundered per tap (per chunk it made the allocator rotate accumulators)
    aw = aw0 + (unsigned)(tap % 3) * WST_BYTES;
  };
  auto read_x = [&](u32x4 (&xf)[4], int ks) {
    const unsigned kx = (unsigned)ks << 6;
    if constexpr (WI == 64) {
      static_assert(srow_d[1] == 16 && srow_d[2] == 32 && srow_d[3] == 48, "64-wide: pixel tiles 16 slab rows apart");
      const unsigned a = ax[0] ^ kx;
      asm volatile("ds_read_b128 %0, %1" : "=v"(xf[0]) : "v"(a));
      asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(xf[1]) : "v"(a));
      asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(xf[2]) : "v"(a));
      asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(xf[3]) : "v"(a));
    } else if constexpr (WI == 32) {
      static_assert(srow_d[1] == 16 && srow_d[3] == srow_d[2] + 16, "32-wide: pixel tiles pair up 16 slab rows apart");
      const unsigned a = ax[0] ^ kx, b = ax[1] ^ kx;
      asm volatile("ds_read_b128 %0, %1" : "=v"(xf[0]) : "v"(a));
      asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(xf[1]) : "v"(a));
      asm volatile("ds_read_b128 %0, %1" : "=v"(xf[2]) : "v"(b));
      asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(xf[3]) : "v"(b));
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(xf[i]) : "v"(ax[i] ^ kx));
    }
  };
  auto read_w = [&](u32x4& wf, int j, int ks) {
    const unsigned a = aw ^ ((unsigned)ks << 6);
    if (j == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(wf) : "v"(a));
    if (j == 1) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(wf) : "v"(a));
    if (j == 2) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(wf) : "v"(a));
    if (j == 3) asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(wf) : "v"(a));
    if (j == 4) asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(wf) : "v"(a));
  };
  f32x4 acc[4][5];
  // One phase: the 20 MFMAs of (xc, w[0..3] + w4c), column by column, and the requests for the NEXT phase's fragments (k-step nks of the addresses in ax / aw).
  // WRITE-AFTER-READ: a ds_read may not target a register that an MFMA issued just before it still has to read — nothing interlocks an LDS return against a queued
  // MFMA's operand fetch.  (Round 6 found out the hard way: fragments re-read into w[j] right behind column j's MFMAs were fine for weeks and corrupted acc[3][0] of
  // the second wave of a SIMD once the window reads became conflict-free and the LDS answered faster; 32 cycles of s_nop in between made it pass.)  So every request
  // goes to a register whose last reader is at least FOUR MFMAs back:
  //   behind column 0: the next phase's window fragments into xn (last read by the previous phase's column 4) and its column-4 weight fragment into w4n (two registers
  //   alternate for column 4); behind column j = 1..4: w[j - 1] (last read by column j - 1).
  // Issue order per phase = [x 4][w4][w0][w1][w2][w3]; LDS returns in order, so with those nine outstanding at a phase's start: column 0 needs the oldest six ->
  // lgkmcnt(3); then five more are issued and columns 1..3 need the oldest of eight -> lgkmcnt(7) (3 without the window requests); column 4's fragment arrived with
  // column 0's wait.  chunk_head (first phase of a chunk: the window fragments were requested LAST, behind barrier X) and a tile's last phase wait for everything.
  auto phase = [&](const u32x4 (&xc)[4], u32x4 (&xn)[4], u32x4 (&w)[4], const u32x4& w4c, u32x4& w4n, int nks, bool next_x, bool next_w, bool chunk_head) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (j == 0) {
        if (chunk_head) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
      } else if (j < 4) {
        if (!next_w) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else if (next_x) asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!ab_nom) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // accumulate IN PLACE (tied operand): left to the compiler the 20 accumulator tuples get renamed from MFMA to MFMA (dst != srcC) on a full register
          // file and the allocator ends up bouncing accumulators or fragment tuples through scratch inside the K loop
          // (the compiler does not know these statements are MFMAs: it inserts no wait states for an accumulator it reads or moves itself.  The build refuses an
          // instance whose K loop touches an accumulator outside an MFMA: theatergen_amd/build.py: check_mfma_loops.)
          if constexpr (sizeof(T) == 2 && std::is_same<T, bf16_t>::value)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(j < 4 ? w[j < 4 ? j : 0] : w4c), "v"(xc[i]));
          else
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(j < 4 ? w[j < 4 ? j : 0] : w4c), "v"(xc[i]));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (j == 0) {
        if (next_x) read_x(xn, nks);
        if (next_w) read_w(w4n, 4, nks);
      } else if (next_w) {
        read_w(w[j - 1], j - 1, nks);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
    int t, sp;
    work_item(v, t, sp);
    const int lbid = t * S + sp;
    const int tile_n = t % p.tiles_n, tile_m = t / p.tiles_n;
    const long m0 = (long)tile_m * BM, n0 = (long)tile_n * BN;
    const int nchunks = nchunks_all - sp * cps < cps ? nchunks_all - sp * cps : cps;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    SLAB_BAR();                                     // S1
    SLAB_BAR();                                     // S2
    u32x4 xa[4], xb[4], wf[4], w4a, w4b;
    set_x(0);
    set_w(0);
    read_x(xa, 0);
    read_w(w4a, 4, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) read_w(wf[j], j, 0);
    for (int cc = 0; cc < nchunks; ++cc) {
      const bool more = cc + 1 < nchunks;
      // per-tap window / weight-stage addresses are recomputed, not hoisted over the chunk loop (and spilled).  (16-wide instance: laundering the weight-stage base
      // too makes the allocator rotate accumulator tuples at the loop's back edge — v_mov of registers the inline-asm MFMAs write: the build's loop check refuses that.)
      if constexpr (WI == 16) asm volatile("" : "+v"(srow0));
      else asm volatile("" : "+v"(srow0), "+v"(aw0));
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        // first phase (k-step 0); the second k-step's fragments come from the same window rows / weight stage
        phase(xa, xb, wf, w4a, w4b, 1, true, true, tap == 0);
        // seam: every read of this K-step's weight stage (and, at tap 8, of the window) has been issued: wait for them, then the workgroup barrier —
        // behind it the loaders refill the stage, and the second phase reads the NEXT K-step's first fragments (its tile has landed)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SLAB_BAR();
        const bool last = !more && tap == 8;
        if (!last) { set_w(tap == 8 ? 0 : tap + 1); if (tap != 8) set_x(tap + 1); }
        __builtin_amdgcn_sched_barrier(0);
        if (tap == 8) {
          // chunk boundary: the loaders are rewriting the window — weight fragments now, pixel fragments behind barrier X
          phase(xb, xa, wf, w4b, w4a, 0, false, !last, false);
          if (more) {
            SLAB_BAR();                             // X
            set_x(0);
            read_x(xa, 0);
          }
        } else {
          phase(xb, xa, wf, w4b, w4a, 0, true, true, false);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // the inline-asm MFMAs' results are read by the epilogue's LDS writes: the wait states the compiler would insert

    // ---- epilogue: fp32 bounce through this wave's 4352 bytes of the slab region
    const int lane_e = lane;
    const int fq_e = lane_e >> 4;
    const int frow_e = ((lane_e & 15) >= 4 && (lane_e & 15) < 12) ? 2 * ((lane_e & 15) - 4) + 1 : ((lane_e & 15) < 4 ? 2 * (lane_e & 15) : 2 * ((lane_e & 15) - 8));   // = fpix: accumulator column -> pixel
    float* scr = reinterpret_cast<float*>(smem) + wave12 * (16 * 68);
    const long mw = m0 + group * 64, nw = n0 + wave_n * 80;
    T* outp = reinterpret_cast<T*>(p.out);
    const T* biasp = reinterpret_cast<const T*>(p.bias);
    const T* bvecp = reinterpret_cast<const T*>(p.bvec);
    const T* resp = reinterpret_cast<const T*>(p.res);
    const float scale = p.out_scale;
    const bool part = S > 1;
    float* wsp = part ? p.ws + (long)lbid * BM * BN : nullptr;
    // GroupNorm partial sums of the OUTPUT (tg_gemm_desc.out_gn_partials; unsplit tiles only): every lane_e sums the stored (rounded) values it writes, per channel
    const bool gn = p.gn_part != nullptr && !part;
    // ... per channel PAIR (GroupNorm groups are even-sized and start on even channels): one v_dot2c per pair and moment, 8 accumulators
    float gs[4], gq[4];
    const unsigned one2 = sizeof(T) == 2 && __builtin_bit_cast(unsigned short, from_f32<T>(1.0f)) == 0x3F80 ? 0x3F803F80u : 0x3C003C00u;
    // PATCH: token of tile row lr = row lr / WI, column x0 + lr % WI of the patch (whole-row tiles: m0 + lr)
    long tok0 = 0;
    int IWe = WI;
    if constexpr (PATCH) {
      IWe = p.in_w;
      const int ppr_e = IWe / WI, tpi_e = (p.in_h / TH) * ppr_e;
      const int img_e = tile_m / tpi_e, rem_e = tile_m - img_e * tpi_e, prow_e = rem_e / ppr_e;
      tok0 = ((long)img_e * p.in_h + prow_e * TH) * IWe + (rem_e - prow_e * ppr_e) * WI;
    }
    auto finish8 = [&](const f32x4& lo, const f32x4& hi, long m, long n, const float (&bias_f)[8]) {
      const long lr_ = m - m0;                       // row of the tile
      if constexpr (PATCH) m = tok0 + (lr_ / WI) * IWe + lr_ % WI;
      if (part) {                                   // fp32 partial in tile-local order: the reduce kernel sums the splits and applies the epilogue
        float* q = wsp + lr_ * BN + (n - n0);
        *reinterpret_cast<f32x4*>(q) = lo;
        *reinterpret_cast<f32x4*>(q + 4) = hi;
        return;
      }
      float x[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { x[e] = lo[e] + bias_f[e]; x[4 + e] = hi[e] + bias_f[4 + e]; }
      if (bvecp != nullptr) {
        const V8 a8 = *reinterpret_cast<const V8*>(bvecp + (m / p.rows_per_batch) * p.ldbvec + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += to_f32<T>(a8[e]);
      }
      if (resp != nullptr) {
        const V8 r8 = *reinterpret_cast<const V8*>(resp + m * p.ldres + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += to_f32<T>(r8[e]);
      }
      V8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(x[e] * scale);
      *reinterpret_cast<V8*>(outp + m * p.ldc + n) = o;
      if (gn) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          const unsigned pr = __builtin_bit_cast(u32x4, o)[e];
          gs[e] = dot2acc<T>(pr, one2, gs[e]);
          gq[e] = dot2acc<T>(pr, pr, gq[e]);
        }
      }
    };
    float* gsum = scr + 1000;                        // [2][40] channel-pair sums of this wave's 64 pixels (the last floats of the wave's 1088: behind both bounce areas)
    {
      // channels 0-63 of the wave's 80: 16 pixels x 64 channels per pass, lane_e -> (row lane_e / 8 + 8 it, 8-channel piece lane_e % 8)
      const int c = lane_e & 7, r0 = lane_e >> 3;
#pragma unroll
      for (int e = 0; e < 4; ++e) { gs[e] = 0.f; gq[e] = 0.f; }
      const long n = nw + c * 8;
      float bias_f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bias_f[e] = 0.f;
      if (biasp != nullptr && !part) {
        const V8 b8 = *reinterpret_cast<const V8*>(biasp + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) bias_f[e] = to_f32<T>(b8[e]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(scr + frow_e * 68 + 16 * j + 4 * fq_e) = acc[i][j];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const f32x4 lo = *reinterpret_cast<const f32x4*>(scr + (it * 8 + r0) * 68 + c * 8);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(scr + (it * 8 + r0) * 68 + c * 8 + 4);
          finish8(lo, hi, mw + i * 16 + it * 8 + r0, n, bias_f);
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (gn) {
        // fold the 8 row groups (r0 = lane_e / 8) through the (free) bounce area: [r0][piece c][4 pair sums | 4 pair sums of squares], row pitch 68 floats; then
        // lane_e L sums column L over the 8 rows in a fixed order (2 ds_write_b128 + 8 ds_read_b32 per lane_e)
        float* red = scr + r0 * 68 + c * 8;
        *reinterpret_cast<f32x4*>(red) = f32x4{gs[0], gs[1], gs[2], gs[3]};
        *reinterpret_cast<f32x4*>(red + 4) = f32x4{gq[0], gq[1], gq[2], gq[3]};
        __builtin_amdgcn_wave_barrier();
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) a += scr[r * 68 + lane_e];
        __builtin_amdgcn_wave_barrier();
        gsum[((lane_e >> 2) & 1) * 40 + (lane_e >> 3) * 4 + (lane_e & 3)] = a;     // column = piece * 8 + kind * 4 + e  ->  gsum[kind][piece * 4 + e]
      }
    }
    {
      // channels 64-79: two pixel tiles per pass = 32 pixels x 16 channels, lane_e -> (row lane_e / 2, 8-channel piece lane_e % 2)
      const int c = lane_e & 1, r = lane_e >> 1;
      const long n = nw + 64 + c * 8;
#pragma unroll
      for (int e = 0; e < 4; ++e) { gs[e] = 0.f; gq[e] = 0.f; }
      float bias_f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bias_f[e] = 0.f;
      if (biasp != nullptr && !part) {
        const V8 b8 = *reinterpret_cast<const V8*>(biasp + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) bias_f[e] = to_f32<T>(b8[e]);
      }
#pragma unroll
      for (int ip = 0; ip < 2; ++ip) {
        *reinterpret_cast<f32x4*>(scr + frow_e * 20 + 4 * fq_e) = acc[2 * ip][4];
        *reinterpret_cast<f32x4*>(scr + (16 + frow_e) * 20 + 4 * fq_e) = acc[2 * ip + 1][4];
        __builtin_amdgcn_wave_barrier();
        const f32x4 lo = *reinterpret_cast<const f32x4*>(scr + r * 20 + c * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(scr + r * 20 + c * 8 + 4);
        finish8(lo, hi, mw + ip * 32 + r, n, bias_f);
        __builtin_amdgcn_wave_barrier();
      }
      if (gn) {
        // 32 rows (lane_e / 2) x 2 pieces: one shuffle folds row pairs, the 16 pair rows go through the bounce area (pitch 20), lane_e L sums column L % 16 over 4 of
        // the 16 rows, two more shuffles join the four quarters
#pragma unroll
        for (int e = 0; e < 4; ++e) { gs[e] += __shfl_xor(gs[e], 2); gq[e] += __shfl_xor(gq[e], 2); }
        if ((lane_e & 2) == 0) {
          float* red = scr + (lane_e >> 2) * 20 + c * 8;
          *reinterpret_cast<f32x4*>(red) = f32x4{gs[0], gs[1], gs[2], gs[3]};
          *reinterpret_cast<f32x4*>(red + 4) = f32x4{gq[0], gq[1], gq[2], gq[3]};
        }
        __builtin_amdgcn_wave_barrier();
        float a = 0.f;
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2) a += scr[((lane_e >> 4) * 4 + r2) * 20 + (lane_e & 15)];
        a += __shfl_xor(a, 16);
        a += __shfl_xor(a, 32);
        if (lane_e < 16) gsum[((lane_e >> 2) & 1) * 40 + 32 + (lane_e >> 3) * 4 + (lane_e & 3)] = a;
        __builtin_amdgcn_wave_barrier();
        // channels -> groups: lane_e g owns group g of the wave's 80 / cpg whole groups; entry [image][64-pixel block][group] is written by this wave alone
        const int cpg = p.gn_cpg, ppg = cpg >> 1;
        if (lane_e < 80 / cpg) {
          float a = 0.f, b = 0.f;
          for (int j = 0; j < ppg; ++j) { a += gsum[lane_e * ppg + j]; b += gsum[40 + lane_e * ppg + j]; }
          const int hw = p.out_h * p.out_w;
          const int img = (int)mw / hw, blk = ((int)mw - img * hw) >> 6;
          const int groups = (int)(p.N / cpg);
          float* o = p.gn_part + ((long)(img * (hw >> 6) + blk) * groups + nw / cpg + lane_e) * 2;
          o[0] = a;
          o[1] = b;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
#undef SLAB_BAR
}

template <typename T, int WI, bool PRO, bool PATCH>
int launch_slab_pp(const tg_gemm_desc* d, GemmParams p, int splits, hipStream_t st) {
  constexpr int BM = 128, TH = BM / WI, SLAB = (TH + 2) * (WI + 2), SJ = (SLAB + 31) / 32;
  constexpr size_t slab = (size_t)SJ * 32 * 128, scratch = 8 * 16 * 68 * 4;
  const size_t lds = (slab > scratch ? slab : scratch) + 3 * (size_t)320 * 128;
  const long tiles_m = d->M / BM, tiles_n = d->N / 320;
  const int nchunks = (d->c0 + (d->a1 ? d->c1 : 0)) / BK;
  p.tiles_n = (int)tiles_n;
  p.full_tiles = 0;
  p.tail_s = splits;
  p.kt_per_split = (nchunks + splits - 1) / splits;
  p.tile_bm = BM; p.tile_bn = 320;
  p.slab_order = ((long)d->N * d->K > (long)d->M * (d->K / 9) && !(p.flags & 8192)) ? 1 : 0;
  long grid = tiles_m * tiles_n * splits;
  if (grid > 256) grid = 256;
  auto k = conv_slab_pp_kernel<T, WI, PRO, PATCH>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr;
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(768), lds, st, p);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

template <typename T>
int launch_slab_pp_dtype(const tg_gemm_desc* d, const GemmParams& p, int splits, hipStream_t st) {
  const bool pro = d->a_coef != nullptr, patch = p.patch_pwl > 0;
  const int pw = patch ? (1 << p.patch_pwl) : d->out_w;                 // tile width: the whole row or one patch
#define TG_SPP_CASE(W) \
  if (pw == W) {                                                                                                                      \
    if (patch) return pro ? launch_slab_pp<T, W, true, true>(d, p, splits, st) : launch_slab_pp<T, W, false, true>(d, p, splits, st);    \
    return pro ? launch_slab_pp<T, W, true, false>(d, p, splits, st) : launch_slab_pp<T, W, false, false>(d, p, splits, st);             \
  }
  TG_SPP_CASE(64)
  TG_SPP_CASE(32)
  TG_SPP_CASE(16)
#undef TG_SPP_CASE
  tg_set_error("tg_gemm conv: no ping-pong slab kernel for tile width %d", pw);
  return TG_ERR_UNSUPPORTED;
}

}  // namespace

// Called by tg_conv_slab.hip (not part of the C ABI) for tiles of one 64 / 32 / 16-wide patch (the whole row, or a patch of a wider map) with 16-byte aligned epilogue operands.
int tg_conv_slab_pp_launch(const tg_gemm_desc* d, const void* params, int splits, void* stream) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == TG_BF16) return launch_slab_pp_dtype<bf16_t>(d, p, splits, st);
  return launch_slab_pp_dtype<f16_t>(d, p, splits, st);
}

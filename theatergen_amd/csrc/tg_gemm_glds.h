// gemm_glds_kernel: the LDS-DMA MFMA GEMM / implicit-GEMM conv kernel template, shared by tg_gemm.hip (plain / conv instances)
// and tg_gemm_ln.hip (the LayerNorm-fused instances).  See tg_gemm.hip for the file-level description.
#pragma once
#include "tg_gemm_common.h"
#include "tg_xattn_epi.h"

namespace {

// ------------------------------------------------------------------------------------------------------------
// Operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR round trip, no
// ds_write pass).  (A first, register-staged generation of this kernel spent 32 KB of ds_write_b128 per 128x128x64 tile,
// ~415 LDS cycles next to 256 cycles of fragment reads and 512 MFMA cycles: the LDS pipe was its bound; it is gone.)  LDS image: unpadded 128-byte rows (the DMA destination is lane-linear), XOR-swizzled in 16-byte slots
// with key = (row >> 1) & 7; the swizzle is applied on the per-lane SOURCE address and again on the fragment read
// (cdna guide rule 21), which makes the 16-lane ds_read_b128 groups conflict-free.  Out-of-range rows / conv padding
// read from a zero page.  Double-buffered: the DMA of tile t+1 is in flight while tile t is multiplied.

// LN = 1 / 2 (tg_gemm_ln.hip; 1: row statistics taken inside the kernel as described below, 2: precomputed by tg_layernorm_stats and read
// from p.ln_rows — no extra work in the K loop; which one pays is a measurement, see profiles/r3_ln_findings.md):
// LN != 0 (tg_gemm_ln.hip: the projections that follow a LayerNorm — attn1 QKV, attn2 to_q, FeedForward GEGLU; K = C, never
// split): LayerNorm is folded into the GEMM so that the normalised tensor and the layernorm launch do not exist.
//   LN(x) W^T = rstd * (x (W * gamma)^T - mean * u) + v,   u[n] = sum_k (W * gamma)[n, k],  v[n] = sum_k beta[k] W[n, k] (+ bias[n])
// W * gamma is packed once in the storage dtype and u is summed from THOSE rounded values, so acc - mean * u is exactly
// sum_k (x[k] - mean) W'[n, k]: no cancellation beyond fp32 accumulation rounding.  mean / rstd come from the A tiles
// this workgroup streams through LDS anyway (a tile's K loop covers whole rows): two threads per row sum x and x^2 of every K-tile with
// v_dot2 (fp32), combined after the loop (var = E[x^2] - mean^2 on the stored, i.e. storage-dtype-rounded values: what nn.LayerNorm reads).
// sum and sum of squares of 8 storage-dtype values by v_dot2 (fp32 accumulate, no conversions)
template <typename T> __device__ __forceinline__ void ln_row_sums(typename Vec<T>::v8 x, float& s, float& q);
template <> __device__ __forceinline__ void ln_row_sums<bf16_t>(bf16x8 x, float& s, float& q) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  const bf16x2 one = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bf16x2 v = {x[2 * i], x[2 * i + 1]};
    s = __builtin_amdgcn_fdot2_f32_bf16(v, one, s, false);
    q = __builtin_amdgcn_fdot2_f32_bf16(v, v, q, false);
  }
}
template <> __device__ __forceinline__ void ln_row_sums<f16_t>(f16x8 x, float& s, float& q) {
  typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
  const f16x2 one = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f16x2 v = {x[2 * i], x[2 * i + 1]};
    s = __builtin_amdgcn_fdot2(v, one, s, false);
    q = __builtin_amdgcn_fdot2(v, v, q, false);
  }
}

// XA = 80 | 160 (tg_xattn_epi.h; LN = 1, 128 x 160 tiles only): the tile is the LayerNorm-folded to_q of two heads of 80 / one head of 160 channels and the
// kernel's output is the cross-attention result O of those heads — W rows are read with bits 2 / 3 of the MFMA row swapped (q comes out as B fragments).
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool CONV, int STAGES, int BKT, int EPI, int LN = 0, int XA = 0>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) __attribute__((amdgpu_waves_per_eu((LN != 0 && STAGES == 3 && BKT == 32) ? 3 : 2)))
void gemm_glds_kernel(GemmParams p) {
  static_assert(XA == 0 || (LN == 1 && BM == 128 && BN == 160 && WAVES_M == 4 && WAVES_N == 1 && BKT == 64 && !CONV && EPI == 0), "XA: the 128 x 160 LayerNorm-folded tile");
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int PF = STAGES - 1;              // K-tiles kept in flight ahead of the one being multiplied
  constexpr int CH = BKT / 8;                 // 16-byte chunks per LDS row (8 at BK = 64, 4 at BK = 32)
  constexpr int RPI = 64 / CH;                // tile rows covered by one 1-KiB DMA instruction (8 or 16)
  constexpr int KSH = CH == 8 ? 1 : 2;        // swizzle key = (row >> KSH) & (CH - 1): conflict-free ds_read_b128 groups
  constexpr int NDMA = BM / (RPI * NW) + BN / (RPI * NW);   // LDS-DMA instructions per wave per K-tile (constant: invalid rows fetch the zero page)
  constexpr int TM = BM / (WAVES_M * 32);
  constexpr int TN = BN / (WAVES_N * 32);
  constexpr int XJ = BM / (RPI * NW);   // DMA instructions per wave per K-tile for the activation tile
  constexpr int WJ = BN / (RPI * NW);
  static_assert(BKT == 64 || BKT == 32, "BK");
  static_assert(BM % (RPI * NW) == 0 && BN % (RPI * NW) == 0, "every wave issues a whole number of DMA instructions per operand tile");
  constexpr int SCW = TN <= 2 ? TN : 2;       // widest column chunk that goes through the epilogue's LDS bounce in one piece (epilogue_tile_lds)
  static_assert((size_t)NW * 32 * (SCW * 32 + 4) * 4 <= (size_t)STAGES * (BM + BN) * BKT * sizeof(T), "epilogue scratch must fit the operand stages");
  typedef typename Vec<T>::v8 V8;
  static_assert(NW % 2 == 0 && XJ >= 1 && WJ >= 1, "tile / wave layout");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sX = reinterpret_cast<T*>(smem);                 // [2][BM][64]
  T* sW = sX + STAGES * BM * BKT;                     // [STAGES][BN][BKT]
  stagger_first_round(p.flags, smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / WAVES_N;
  const int wave_n = wave % WAVES_N;
  // work item -> (tile, K range): the first full_tiles blocks compute whole tiles (XCD-chunked order); the tail tiles
  // are cut tail_s ways along K so that the last, partially filled round of the grid is spread over all CUs
  int lbid, split = 0, part = -1;
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
  const int nkt_total = (int)((p.K + BKT - 1) / BKT);
  const int kt_begin = part >= 0 ? split * p.kt_per_split : 0;
  int kt_end = part >= 0 ? kt_begin + p.kt_per_split : nkt_total;
  if (kt_end > nkt_total) kt_end = nkt_total;
  const int nkt = kt_end - kt_begin;
  // (K ROTATION — every work item starting its K walk at another tile so that the lockstep workgroups of a single-round launch do not ask the same few L2
  // channels for the same K offset — was built and measured in round 5: the operand-pitch effect is real (profiles/r5_operand_pitch.txt: 4096 x 1280 x 5120
  // 594 TF, 718-751 TF with the row pitch padded by 64 / 192 elements) but the rotation loses more L2 locality than it wins: 16384 x 640 x 2560 70.2 -> 80.1 us.)

  // DMA lane geometry: instruction q covers tile rows [RPI*q, RPI*q + RPI); lane -> (row RPI*q + lane/CH, slot lane%CH)
  const int lrow = lane / CH;
  const int slot = lane & (CH - 1);
  // key of row RPI*q + lrow, q = j*NW + wave: BK=64: ((8q + lrow) >> 1) & 7 = (4(wave&1) + lane/16) & 7;  BK=32: (lane/16) & 3
  const int wkey = CH == 8 ? ((4 * (wave & 1) + (lane >> 4)) & 7) : ((lane >> 4) & 3);
  const int chunk = slot ^ wkey;                          // global 16-byte chunk this lane fetches into its slot

  const T* A0 = reinterpret_cast<const T*>(p.a0);
  const T* A1 = reinterpret_cast<const T*>(p.a1);
  const T* Wp = reinterpret_cast<const T*>(p.w);
  const T* zero = reinterpret_cast<const T*>(tg_zero_page);
  const int ctot = p.c0 + p.c1;

  long xbase[XJ], xrow[XJ];
  int x_oy[XJ], x_ox[XJ], x_ob[XJ];
  bool x_ok[XJ];
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const long m = m0 + (j * NW + wave) * RPI + lrow;
    x_ok[j] = m < p.M;
    xbase[j] = CONV ? m * p.c0 : m * p.lda;         // (plain: lda = c0 unless the caller padded the rows; the two-source form keeps pitch = channel count)
    xrow[j] = m;
    if (!CONV && p.a_rpb > 0) { const long bb = m / p.a_rpb; xbase[j] = bb * p.a_bs + (m - bb * p.a_rpb) * p.c0; }
    if (CONV) {
      const long mm = x_ok[j] ? m : 0;
      const int hw = p.out_h * p.out_w;
      x_ob[j] = (int)(mm / hw);
      const int r = (int)(mm - (long)x_ob[j] * hw);
      x_oy[j] = r / p.out_w;
      x_ox[j] = r - x_oy[j] * p.out_w;
    }
  }
  const T* wrow[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) {
    const long n = n0 + (j * NW + wave) * RPI + lrow;
    wrow[j] = n < p.N ? Wp + n * (CONV ? p.K : p.ldw) : nullptr;
  }

  auto dma = [&](const T* src, T* lds_row_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_row_base, 16, 0, 0);
  };

  // FAST DMA ADDRESSING (round 6): a plain single-source GEMM tile that lies wholly inside M x N with K a multiple of the K-tile needs no per-request pointer
  // arithmetic — the source of a request is a wave-uniform base (operand pointer + the K-tile's offset: SGPR pair, scalar add) plus this lane's constant 32-bit byte
  // offset (its row and swizzled chunk).  The generic path below forms a 64-bit pointer and a zero-page select per request: 45 of the 66 VALU instructions of a
  // 128 x 160 K-tile, and every VALU instruction takes matrix-pipe time with it (profiles/r6_mfma_exp_overlap.json).
  const unsigned lds_sx = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned lds_sw = lds_sx + (unsigned)(STAGES * BM * BKT * sizeof(T));
  const bool fastdma = !CONV && A1 == nullptr && p.a_rpb <= 0 && p.K % BKT == 0 && m0 + BM <= p.M && n0 + BN <= p.N &&
                       (long)p.N * p.ldw * (long)sizeof(T) < (1L << 31) && (long)p.M * p.lda * (long)sizeof(T) < (1L << 31);
  unsigned voffw[WJ], voffx[XJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) voffw[j] = (unsigned)(((n0 + (j * NW + wave) * RPI + lrow) * p.ldw + chunk * 8) * (long)sizeof(T));
#pragma unroll
  for (int j = 0; j < XJ; ++j) voffx[j] = (unsigned)(((m0 + (j * NW + wave) * RPI + lrow) * p.lda + chunk * 8) * (long)sizeof(T));
  auto dma_s = [&](const T* base, unsigned voff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_byte_addr) : "memory");
  };

  auto issue_tile = [&](int kt, int buf) {
    const long k0 = (long)kt * BKT;
    if constexpr (!CONV) {
      if (fastdma) {
        const T* wb = Wp + k0;
        const T* xb = A0 + k0;
#pragma unroll
        for (int j = 0; j < WJ; ++j) dma_s(wb, voffw[j], lds_sw + (unsigned)((buf * BN * BKT + (j * NW + wave) * RPI * BKT) * sizeof(T)));
#pragma unroll
        for (int j = 0; j < XJ; ++j) dma_s(xb, voffx[j], lds_sx + (unsigned)((buf * BM * BKT + (j * NW + wave) * RPI * BKT) * sizeof(T)));
        return;
      }
    }
    const long kc = k0 + chunk * 8;
    const bool kok = kc < p.K;
    T* dx = sX + buf * BM * BKT;
    T* dw = sW + buf * BN * BKT;
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const T* src = (kok && wrow[j] != nullptr) ? wrow[j] + kc : zero;
      dma(src, dw + (j * NW + wave) * RPI * BKT);
    }
    if (!CONV) {
      const T* base = A0;
      long pitch = p.c0;
      long kk = kc;
      const bool second = A1 != nullptr && k0 >= p.c0;
      if (second) { base = A1; pitch = p.c1; kk = kc - p.c0; }
#pragma unroll
      for (int j = 0; j < XJ; ++j) {
        const long off = second ? xrow[j] * pitch : xbase[j];
        const T* src = (kok && x_ok[j]) ? base + off + kk : zero;
        dma(src, dx + (j * NW + wave) * RPI * BKT);
      }
    } else {
      const int tap = (int)(k0 / ctot);
      int cc = (int)(k0 - (long)tap * ctot);
      const int ky = tap / 3, kx = tap - ky * 3;
      const T* base = A0;
      int pitch = p.c0;
      if (cc >= p.c0) { base = A1; pitch = p.c1; cc -= p.c0; }
      cc += chunk * 8;
#pragma unroll
      for (int j = 0; j < XJ; ++j) {
        int iy, ix;
        bool ok = x_ok[j];
        if (!p.upsample) {
          iy = x_oy[j] * p.stride + ky - p.pad_lo;
          ix = x_ox[j] * p.stride + kx - p.pad_lo;
          ok = ok && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
        } else {
          const int uy = x_oy[j] + ky - 1, ux = x_ox[j] + kx - 1;
          ok = ok && uy >= 0 && uy < 2 * p.in_h && ux >= 0 && ux < 2 * p.in_w;
          iy = uy >> 1;
          ix = ux >> 1;
        }
        const T* src = ok ? base + ((long)(x_ob[j] * p.in_h + iy) * p.in_w + ix) * pitch + cc : zero;
        dma(src, dx + (j * NW + wave) * RPI * BKT);
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // LN: thread -> (tile row tid / 2, half tid & 1 of the row's K-tile); SCH 16-byte slots per thread and K-tile.  The slot order is
  // rotated by the row so that the 16 lanes of one ds_read_b128 phase hit 16 different 4-bank groups.
  static_assert(LN == 0 || (BM * 2 == NW * 64 && !CONV && TN <= 5), "LN: two threads per tile row, plain GEMM");
  constexpr int SCH = CH / 2;
  const int srow = tid >> 1, shalf = tid & 1;
  const int srot = CH == 8 ? (srow >> 1) : (srow >> 2);
  float ln_s = 0.f, ln_q = 0.f;

  // big wave tiles: fragments are read per k-step (register budget).  The 1 x 5 wave tile of the 128 x 160 kernels (round 5) still reads a whole K-tile's
  // 24 fragments up front (96 VGPRs + 80 accumulators = 212 .. 239 VGPRs incl. the LayerNorm fold, no scratch): per k-step hipcc emits read-2 / lgkmcnt(0) /
  // mfma groups on ONE recycled register quad — ten exposed LDS latencies per K-tile with two waves per SIMD (4096 x 1280 x 5120: 79 -> 72 us)
  constexpr bool BIGW = TM * TN > 5;
  // EARLY REFILL (2 stages, all fragments of a K-tile read into registers up front): a stage is dead as soon as every wave
  // has its 16 fragments, i.e. half a K-tile before the next one starts — it is refilled right then with the tile AFTER
  // next.  Two K-tiles are in flight with two 32 KB stages; the K-tile period was one DMA round trip (~1800 cycles against
  // 1024 of MFMA work for the two resident blocks) and a third stage does not fit next to a second block.
  constexpr bool ER = STAGES == 2 && !BIGW;
  if (nkt > 0) {
    // counted waits: a wave only waits until the NEXT tile's DMA has landed (vmcnt(NDMA) = one younger tile may stay in
    // flight; LDS-DMA completes in issue order); raw s_barrier, because __syncthreads() would drain vmcnt to 0.
    if constexpr (ER) {
      issue_tile(kt_begin + (0), 0);
      if (nkt > 1) issue_tile(kt_begin + (1), 1);
      if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int s = 0; s < PF; ++s)
        if (s < nkt) issue_tile(kt_begin + (s), s);
      if (PF >= 2 && nkt >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const int l31 = lane & 31;
    const int hi = lane >> 5;
    const int rkey = (l31 >> KSH) & (CH - 1);
    const int l31w = XA != 0 ? xa_swap23(l31) : l31;            // XA: MFMA row i of a W tile = weight row swap23(i) (same 16-lane bank groups: conflict-free)
    const int rkeyw = (l31w >> KSH) & (CH - 1);
    int buf = 0;
    for (int it = 0; it < nkt; ++it) {
      const T* bx = sX + buf * BM * BKT + (wave_m * TM * 32 + l31) * BKT;
      const T* bw = sW + buf * BN * BKT + (wave_n * TN * 32 + l31w) * BKT;
      // all fragment reads of the K-tile first (16 ds_read_b128 = 64 VGPRs at 2x2 tiles), then one uninterrupted
      // MFMA chain: the compiler's counted lgkmcnt waits then expose the LDS latency once per tile instead of once
      // per k-step (it otherwise emits read-4 / wait-all / mfma-4 groups and the matrix pipe idles ~50 % per wave).
      V8 xf[BKT / 16][TM], wf[BKT / 16][TN];
      if constexpr (!BIGW) {
#pragma unroll
        for (int ks = 0; ks < BKT / 16; ++ks) {
          const int so = ((2 * ks + hi) ^ rkey) * 8, sow = ((2 * ks + hi) ^ rkeyw) * 8;
#pragma unroll
          for (int i = 0; i < TM; ++i) xf[ks][i] = *reinterpret_cast<const V8*>(bx + i * 32 * BKT + so);
#pragma unroll
          for (int j = 0; j < TN; ++j) wf[ks][j] = *reinterpret_cast<const V8*>(bw + j * 32 * BKT + sow);
        }
      }
      // LN row sums: with the early refill the stage dies in the middle of this iteration, so the thread's slots are read up here
      // next to the fragments; otherwise (3 stages) they are read BEHIND the MFMA chain, when the fragment registers are dead —
      // the three-workgroups-per-CU variant has 168 registers per lane and not one to spare
      V8 sx[SCH];
      const T* srp = sX + buf * BM * BKT + srow * BKT + shalf * (SCH * 8);
      if constexpr (LN == 1 && ER) {
#pragma unroll
        for (int c = 0; c < SCH; ++c) sx[c] = *reinterpret_cast<const V8*>(srp + (((c + srot) & (SCH - 1)) << 3));
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!ER) {
        // the next tile's DMA addresses are computed / issued while the fragment reads are in flight
        if (it + PF < nkt) {
          int nb = buf + PF;
          if (nb >= STAGES) nb -= STAGES;
          issue_tile(kt_begin + (it + PF), nb);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (p.flags & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < BKT / 16; ++ks) {
        if constexpr (BIGW) {
          const int so = ((2 * ks + hi) ^ rkey) * 8, sow = ((2 * ks + hi) ^ rkeyw) * 8;
#pragma unroll
          for (int i = 0; i < TM; ++i) xf[ks][i] = *reinterpret_cast<const V8*>(bx + i * 32 * BKT + so);
#pragma unroll
          for (int j = 0; j < TN; ++j) wf[ks][j] = *reinterpret_cast<const V8*>(bw + j * 32 * BKT + sow);
        }
        if constexpr (ER) {
          if (ks == BKT / 32) {
            // half of the chain is issued: by now every fragment has landed in registers -> the stage is free
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (it + 2 < nkt) issue_tile(kt_begin + (it + 2), buf);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(wf[ks][j], xf[ks][i], acc[i][j]);
      }
      if (p.flags & 2) __builtin_amdgcn_s_setprio(0);
      if constexpr (LN == 1) {
        if constexpr (!ER) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < SCH; ++c) sx[c] = *reinterpret_cast<const V8*>(srp + (((c + srot) & (SCH - 1)) << 3));
        }
#pragma unroll
        for (int c = 0; c < SCH; ++c) ln_row_sums<T>(sx[c], ln_s, ln_q);
      }
      // keep the MFMA chain ABOVE the wait: an asm "memory" clobber does not order register-only MFMAs, and hipcc
      // otherwise hoists `s_waitcnt vmcnt(0); s_barrier` in front of them, exposing the whole DMA latency per tile
      __builtin_amdgcn_sched_barrier(0);
      // tile it+1 must have landed; a younger tile (if issued) may stay in flight
      if ((ER || PF >= 2) && it + 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      buf = buf + 1 == STAGES ? 0 : buf + 1;
    }
  }

  if constexpr (LN != 0) {
    // row statistics -> (a, b) = (rstd, -rstd * mean) and this tile's u[n] into LDS behind the epilogue scratch (the operand stages
    // are dead); then, still in accumulator layout (lane & 31 = row: a, b are per-lane scalars; a register quad = 4 consecutive
    // columns: one broadcast ds_read_b128 of u), acc <- a * acc + b * u = rstd * (acc - mean * u), in place
    static_assert((size_t)(NW * 32 * (SCW * 32 + 4) + 2 * BM + BN) * 4 <= (size_t)STAGES * (BM + BN) * BKT * sizeof(T), "LN: row statistics must fit the operand stages");
    float* lnrow = reinterpret_cast<float*>(smem) + NW * 32 * (SCW * 32 + 4);
    float* lnu = lnrow + 2 * BM;
    if constexpr (LN == 1) {
      const float s_all = ln_s + __shfl_xor(ln_s, 1, 64), q_all = ln_q + __shfl_xor(ln_q, 1, 64);
      const float inv_k = 1.0f / (float)p.K;
      const float mean = s_all * inv_k;
      float var = fmaxf(__builtin_fmaf(-mean, mean, q_all * inv_k), 0.f);
      // E[x^2] - mean^2 cancels when |mean| >> std: its absolute error is ~1e-6 * mean^2, so below var = 1e-4 * mean^2
      // (|mean| / std > 100) the row's variance is RE-TAKEN centred, from global memory, by the row's two threads (rare rows only: the
      // fast path of every other row is untouched; tg_layernorm takes the same two-pass form)
      if (var < 1e-4f * mean * mean && m0 + srow < p.M) {
        const long m = m0 + srow;
        long off = m * p.lda;
        if (p.a_rpb > 0) { const long bb = m / p.a_rpb; off = bb * p.a_bs + (m - bb * p.a_rpb) * p.c0; }
        const T* xr = A0 + off;
        float c2 = 0.f;
        for (long k = shalf * 8; k < p.K; k += 16) {
          const V8 v = *reinterpret_cast<const V8*>(xr + k);
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float dlt = to_f32<T>(v[e]) - mean; c2 = __builtin_fmaf(dlt, dlt, c2); }
        }
        c2 += __shfl_xor(c2, 1, 64);
        var = c2 * inv_k;
      }
      const float rstd = __builtin_amdgcn_rsqf(var + p.ln_eps);
      if (shalf == 0) {
        lnrow[2 * srow] = rstd;
        lnrow[2 * srow + 1] = -rstd * mean;
      }
    } else {
      if (tid < BM) {
        const long m = m0 + tid;
        float2 ab = make_float2(1.f, 0.f);
        if (m < p.M) ab = *reinterpret_cast<const float2*>(p.ln_rows + 2 * m);
        *reinterpret_cast<float2*>(lnrow + 2 * tid) = ab;
      }
    }
    if (tid < BN / 4) {
      const long n4 = n0 + 4 * tid;
      f32x4 u4 = {0.f, 0.f, 0.f, 0.f};
      if (n4 < p.N) u4 = *reinterpret_cast<const f32x4*>(p.ln_u + n4);
      *reinterpret_cast<f32x4*>(lnu + 4 * tid) = u4;
    }
    __syncthreads();
    {
      const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int lr = (wave_m * TM + i) * 32 + l31;
        const float a = lnrow[2 * lr], b = lnrow[2 * lr + 1];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            // (XA: the W rows are swapped, register quad g of lane half hi holds channels 16 (g >> 1) + 8 hi + 4 (g & 1) .. + 3 of the 32-column tile)
            const f32x4 u4 = *reinterpret_cast<const f32x4*>(lnu + (wave_n * TN + j) * 32 + (XA != 0 ? 16 * (g >> 1) + 8 * hi + 4 * (g & 1) : 8 * g + 4 * hi));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = __builtin_fmaf(acc[i][j][4 * g + e], a, b * u4[e]);
          }
      }
    }
    if constexpr (XA != 0) {
      // cross-attention of the tile's heads on the q the accumulators hold; O^T comes back in `acc` (standard layout) and leaves as a plain tile
      xattn_epilogue<T, XA>(p, acc, smem, wave, lane, m0, n0, tile_n);
      GemmParams po = p;
      po.bias = nullptr; po.bvec = nullptr; po.res = nullptr; po.n_split = 0; po.out_scale = 1.0f; po.act = TG_ACT_NONE; po.geglu = 0;
      epilogue_tile_lds<T, TM, TN, EPI, false>(po, acc, m0 + wave_m * TM * 32, n0 + wave_n * TN * 32, lane,
                                               reinterpret_cast<float*>(smem) + wave * (32 * (SCW * 32 + 4)), -1, m0, n0);
    } else {
      epilogue_tile_lds<T, TM, TN, EPI, true>(p, acc, m0 + wave_m * TM * 32, n0 + wave_n * TN * 32, lane,
                                              reinterpret_cast<float*>(smem) + wave * (32 * (SCW * 32 + 4)), part, m0, n0);
    }
  } else {
    epilogue_tile_lds<T, TM, TN, EPI>(p, acc, m0 + wave_m * TM * 32, n0 + wave_n * TN * 32, lane,
                                     reinterpret_cast<float*>(smem) + wave * (32 * (SCW * 32 + 4)), part, m0, n0);
  }
}

}  // namespace

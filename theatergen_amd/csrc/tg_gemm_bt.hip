// Big-tile MFMA GEMM for gfx950 ("bt" kernels; include/theatergen_hip.h: tg_gemm, selected by the planner in tg_gemm.hip).
//
// Why a second kernel family.  The 128x128 kernels of tg_gemm.hip move 32 KB of operands from L2 into LDS per 2 * 128 * 128 * 64
// FLOP: at the MFMA peak that is 64 B/clk per CU = 39 TB/s over the chip, MORE than the 34.5 TB/s the eight L2s deliver
// (MI355X_MICROARCH.md) — the structure is capped by L2 -> LDS traffic long before the matrix pipe, and round 1 measured exactly
// that (0.55 .. 0.9 PF, K-tile period 1800 cycles against 1024 of MFMA work).  Here ONE 8-wave workgroup per CU owns a
// 256 x 320 (or 128 x 320 / 256 x 256) output tile: 2.2x fewer operand bytes per FLOP, all N = 320-multiples of the SD UNets are
// whole tiles (no 17 % ragged third tile at N = 320), and M = 65536 / 256 = 256 row tiles = one per CU.
//
// Structure (plain HIP + LDS-DMA, 512 threads = 4 x 2 waves of (BM/4) x (BN/2) outputs, acc 160 registers at 256 x 320):
//   * operands HBM/L2 -> LDS by global_load_lds_dwordx4 into two 72 KB stages, 128-byte rows, the same XOR swizzle as the
//     128x128 kernels (applied on the source address and on the fragment read; conflict-free ds_read_b128);
//   * a K-tile is 4 k-steps of TM x TN MFMAs (40 at 256x320 = 1280 matrix-pipe cycles per wave, 2560 per SIMD): long enough
//     that ONE tile of prefetch hides the whole DMA round trip, so two stages suffice.  Fragments are double-buffered in
//     registers: the reads of k-step s+1 are in flight under the MFMAs of k-step s (LDS time 34 % of MFMA time at 256x320);
//   * ONE barrier per K-tile, placed before the LAST k-step: by then every wave has its last fragments of the current stage in
//     registers (lgkmcnt(0)) and the next tile has landed (vmcnt(0), issued a whole K-tile earlier), so behind the barrier the
//     stage just read is refilled with the tile after next, and the first fragments of the next tile are read under the last
//     MFMAs — no exposed LDS latency at the tile seam;
//   * persistent: a workgroup walks its tiles (XCD-chunked order: neighbours share A rows through one L2); the first K-tile of
//     the NEXT output tile is requested before the epilogue of the current one starts (into the stage the epilogue's LDS
//     bounce does not use), so only the first tile of a workgroup pays the cold operand fetch;
//   * epilogue: the shared LDS-transposed one (tg_gemm_common.h) in 64-column chunks: 16-byte stores on whole 128-byte rows.
#include "tg_gemm_common.h"

namespace {

template <typename T, int BM, int BN, int WM, int WN, int EPI, bool REGEPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void bt_gemm_kernel(GemmParams p) {
  constexpr int NW = 8;
  static_assert(WM * WN == NW, "8 waves");
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int XJ = BM / 64, WJ = BN / 64;            // LDS-DMA instructions per wave per K-tile (8 rows x 128 B each)
  // elements per stage: X rows [0, BM), W rows [BM, BM + BN); the stage PITCH also covers the epilogue's LDS bounce (8 waves x
  // 32 x 68 floats), which runs in the lower stage while the next tile's first K-tile lands in the upper one
  constexpr int SCRATCH = NW * 32 * 68 * 4 / (int)sizeof(T);
  constexpr int STAGE = (BM + BN) * BK > SCRATCH ? (BM + BN) * BK : SCRATCH;
  static_assert(BM % 64 == 0 && BN % 64 == 0, "tile rows are dealt to the 8 waves in groups of 8");
  typedef typename Vec<T>::v8 V8;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sbase = reinterpret_cast<T*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / WN, wave_n = wave % WN;
  const int l31 = lane & 31, hi = lane >> 5;
  const int rkey = (l31 >> 1) & 7;

  // DMA lane geometry (as in tg_gemm.hip): instruction q covers tile rows [8q, 8q + 8); lane -> (row 8q + lane / 8, slot lane % 8);
  // the 16-byte chunk fetched into a slot is slot ^ key(row), key(row) = (row >> 1) & 7 = (4 (q & 1) + lane / 16) & 7, q & 1 = wave & 1
  const int lrow = lane >> 3;
  const int chunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);

  const T* A0 = reinterpret_cast<const T*>(p.a0);
  const T* A1 = reinterpret_cast<const T*>(p.a1);
  const T* Wp = reinterpret_cast<const T*>(p.w);
  const T* zero = reinterpret_cast<const T*>(tg_zero_page);
  const int nkt = (int)((p.K + BK - 1) / BK);
  const int tiles_m = (int)((p.M + BM - 1) / BM);
  const int ntiles = tiles_m * p.tiles_n;

  // LDS-DMA by inline asm: hipcc's waitcnt pass puts `s_waitcnt vmcnt(0)` in front of every ds_read that follows a
  // global_load_lds BUILTIN in program order (it cannot tell that the fragment reads never touch the stage being filled), which
  // would expose the whole DMA round trip once per k-step.  An asm statement is invisible to that pass: all operand waits of
  // the K loop are the counted ones written below (guide 5.7: count your own queue; M0 is written in the statement that uses it).
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto dma = [&](const T* src, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_byte_addr)
                 : "memory");
  };

  // per-tile operand row pointers: the lane's 16-byte chunk of K-tile 0 of its row (or the zero page for rows outside the
  // problem, which must not advance with k): one 64-bit add per DMA instruction in the loop
  const T* xptr[XJ];
  const T* wptr[WJ];
  unsigned xstep[XJ], wstep[WJ];            // 1 = the row advances with k, 0 = zero page
  auto setup_tile = [&](long m0, long n0) {
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const long m = m0 + (j * NW + wave) * 8 + lrow;
      long off = m * p.c0;
      if (p.a_rpb > 0) { const long bb = m / p.a_rpb; off = bb * p.a_bs + (m - bb * p.a_rpb) * p.c0; }
      const bool ok = m < p.M;
      xptr[j] = ok ? A0 + off + chunk * 8 : zero;
      xstep[j] = ok ? 1u : 0u;
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const long n = n0 + (j * NW + wave) * 8 + lrow;
      const bool ok = n < p.N;
      wptr[j] = ok ? Wp + n * p.K + chunk * 8 : zero;
      wstep[j] = ok ? 1u : 0u;
    }
  };
  auto issue_tile = [&](int kt, int stage) {
    const unsigned k0 = (unsigned)kt * BK;                       // elements; K % 64 == 0 and a single A source (planner rule)
    const unsigned dx = lds0 + (unsigned)(stage * STAGE + wave * 8 * BK) * (unsigned)sizeof(T);
    const unsigned dw = dx + (unsigned)(BM * BK) * (unsigned)sizeof(T);
#pragma unroll
    for (int j = 0; j < WJ; ++j) dma(wptr[j] + k0 * wstep[j], dw + (unsigned)(j * NW * 8 * BK) * (unsigned)sizeof(T));
#pragma unroll
    for (int j = 0; j < XJ; ++j) dma(xptr[j] + k0 * xstep[j], dx + (unsigned)(j * NW * 8 * BK) * (unsigned)sizeof(T));
  };
  // Fragment reads by inline asm as well: hipcc sinks plain LDS loads down to their first use (lgkmcnt(1..2) ladders with the
  // LDS latency exposed between the MFMAs of a k-step); an asm volatile ds_read stays where it is written, so the reads of
  // k-step s + 1 really are in flight under the MFMAs of k-step s.  The matching counted lgkmcnt waits are written by hand
  // (LDS operations return in order: after NF newer reads were issued, lgkmcnt(NF) means the previous set has arrived).
  constexpr int NF = TM + TN;
  unsigned so[4];                                   // per-lane byte offset of k-step ks inside a 128-byte row (swizzled)
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) so[ks] = (unsigned)(((2 * ks + hi) ^ rkey) * 16);
  const unsigned fx0 = lds0 + (unsigned)((wave_m * TM * 32 + l31) * BK) * (unsigned)sizeof(T);
  const unsigned fw0 = lds0 + (unsigned)((BM + wave_n * TN * 32 + l31) * BK) * (unsigned)sizeof(T);
  constexpr unsigned STAGE_BYTES = (unsigned)STAGE * (unsigned)sizeof(T);
  constexpr int TSTEP = 32 * BK * (int)sizeof(T);   // bytes between two 32-row tiles of a wave tile (immediate offset)
  auto read_frags = [&](u32x4 (&xf)[TM], u32x4 (&wf)[TN], int stage, int ks) {
    const unsigned ax = fx0 + (unsigned)stage * STAGE_BYTES + so[ks];
    const unsigned aw = fw0 + (unsigned)stage * STAGE_BYTES + so[ks];
    if constexpr (TM >= 1) asm volatile("ds_read_b128 %0, %1" : "=v"(xf[0]) : "v"(ax));
    if constexpr (TM >= 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[TM >= 2 ? 1 : 0]) : "v"(ax), "n"(1 * TSTEP));
    asm volatile("ds_read_b128 %0, %1" : "=v"(wf[0]) : "v"(aw));
    if constexpr (TN >= 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[TN >= 2 ? 1 : 0]) : "v"(aw), "n"(1 * TSTEP));
    if constexpr (TN >= 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[TN >= 3 ? 2 : 0]) : "v"(aw), "n"(2 * TSTEP));
    if constexpr (TN >= 4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[TN >= 4 ? 3 : 0]) : "v"(aw), "n"(3 * TSTEP));
    if constexpr (TN >= 5) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[TN >= 5 ? 4 : 0]) : "v"(aw), "n"(4 * TSTEP));
    static_assert(TM <= 2 && TN <= 5, "fragment reads are written out for wave tiles up to 64 x 160");
  };
  auto mfmas = [&](f32x16 (&acc)[TM][TN], const u32x4 (&xf)[TM], const u32x4 (&wf)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(__builtin_bit_cast(V8, wf[j]), __builtin_bit_cast(V8, xf[i]), acc[i][j]);
  };
  // wait until at most N LDS reads of this wave are outstanding; fences the MFMAs below it (guide 5.4 rule 18)
#define BT_LGKM(N)                                         \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)

  // reg16 epilogue (no LDS): BOTH first K-tiles of the next output tile are requested before the epilogue of the current one;
  // the LDS-bounce epilogue (per-batch vector adds, unaligned operands) leaves only the upper stage for that
  // (REGEPI is a template parameter: with both epilogues in one kernel the register allocator spills around the K loop)
  constexpr bool reg_epi = REGEPI;
  int par = 0;                              // stage holding K-tile 0 of the current output tile
  int pre = 0;                              // K-tiles of the current output tile already requested (0, 1 or 2)
  for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
    const int lbid = xcd_chunked_block_id(v, ntiles);
    const int tile_n = lbid % p.tiles_n, tile_m = lbid / p.tiles_n;
    const long m0 = (long)tile_m * BM, n0 = (long)tile_n * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (pre == 0) {
      setup_tile(m0, n0);
      issue_tile(0, par);
    }
    if (pre < 2 && nkt > 1) issue_tile(1, par ^ 1);
    // K-tile 0 must have landed; K-tile 1 (XJ + WJ instructions per wave, issued last) may stay in flight
    if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XJ + WJ) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    u32x4 xa[TM], wa[TN], xb[TM], wb[TN];
    read_frags(xa, wa, par, 0);
    for (int t = 0; t < nkt; ++t) {
      const int st = (t + par) & 1;
      read_frags(xb, wb, st, 1);                 // k-step 0 on fragments a; the reads of k-step 1 are in flight under it
      BT_LGKM(NF);
      mfmas(acc, xa, wa);
      __builtin_amdgcn_sched_barrier(0);
      read_frags(xa, wa, st, 2);
      BT_LGKM(NF);
      mfmas(acc, xb, wb);
      __builtin_amdgcn_sched_barrier(0);
      read_frags(xb, wb, st, 3);
      BT_LGKM(NF);
      mfmas(acc, xa, wa);
      __builtin_amdgcn_sched_barrier(0);
      // seam: my reads of stage st are complete (and the fragments of k-step 3 are in registers), the next tile — the only
      // DMA in flight, requested one K-tile ago — has landed -> ONE barrier per K-tile
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (t + 1 < nkt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < nkt) read_frags(xa, wa, st ^ 1, 0);          // first fragments of the next tile under the last MFMAs
      if (t + 2 < nkt) issue_tile(t + 2, st);                   // refill the stage just read with the tile after next
      __builtin_amdgcn_sched_barrier(0);
      mfmas(acc, xb, wb);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- output-tile seam: every wave is past the last barrier of the K loop, i.e. all fragment reads are done and the
    // whole LDS is free.  Request the first K-tile(s) of the next output tile now: they land under the epilogue.
    const int vn = v + gridDim.x;
    const long pm0 = m0, pn0 = n0;
    pre = 0;
    if (vn < ntiles) {
      const int lb2 = xcd_chunked_block_id(vn, ntiles);
      setup_tile((long)(lb2 / p.tiles_n) * BM, (long)(lb2 % p.tiles_n) * BN);
      if (reg_epi) {
        par = 0;
        issue_tile(0, 0);
        pre = 1;
        if (nkt > 1) { issue_tile(1, 1); pre = 2; }
      } else {
        par = 1;                                               // the LDS bounce runs in the lower stage
        issue_tile(0, 1);
        pre = 1;
      }
    }
    if constexpr (reg_epi) {
      epilogue_tile_reg16<T, TM, TN, EPI>(p, acc, pm0 + wave_m * TM * 32, pn0 + wave_n * TN * 32, lane, pm0, pn0);
    } else {
      epilogue_tile_lds<T, TM, TN, EPI>(p, acc, pm0 + wave_m * TM * 32, pn0 + wave_n * TN * 32, lane,
                                       reinterpret_cast<float*>(smem) + wave * (32 * 68), -1, pm0, pn0);
      // the next tile's K-tile 1 goes into the lower stage: every wave must be out of its bounce first
      __builtin_amdgcn_s_barrier();
    }
  }
}

#undef BT_LGKM

template <typename T, int BM, int BN, int WM, int WN, int EPI, bool REGEPI>
void launch_bt_variant(const GemmParams& p, long grid, size_t lds, hipStream_t st) {
  auto k = bt_gemm_kernel<T, BM, BN, WM, WN, EPI, REGEPI>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr;
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(512), lds, st, p);
}
template <typename T, int BM, int BN, int WM, int WN, int EPI>
void launch_bt_kernel(const GemmParams& p, long grid, size_t lds, hipStream_t st) {
  // The register-only 16-byte epilogue (reg16) is used for the fused GEGLU only.  Measured on MI355X (scripts/dev_bt_bench.py,
  // rotating operands): GEGLU 65536x2560x320 187.6 us with reg16 vs 197.3 with the LDS bounce, 16384x5120x640 156.4 vs 161.9;
  // but LINEAR epilogues lose badly without the bounce (65536x320x320 + residual 66.9 vs 39.2 us, 16384x640x640 43.8 vs 25.6):
  // 32 contiguous bytes per token row and instruction (two lanes per row) cost twice the L2 / TA line accesses of the
  // bounce's whole 128-byte rows, for the residual loads as well as for the stores — row-contiguous access wins there.
  if constexpr (EPI == 2) {
    if (p.epi_lds && p.bvec == nullptr && !(p.flags & 4)) { launch_bt_variant<T, BM, BN, WM, WN, EPI, true>(p, grid, lds, st); return; }
  }
  launch_bt_variant<T, BM, BN, WM, WN, EPI, false>(p, grid, lds, st);
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_bt(const tg_gemm_desc* d, GemmParams p, hipStream_t st) {
  const size_t stage = (size_t)(BM + BN) * BK * sizeof(T), scratch = (size_t)8 * 32 * 68 * 4;
  const size_t lds = 2 * (stage > scratch ? stage : scratch);
  const long tiles_m = (d->M + BM - 1) / BM, tiles_n = (d->N + BN - 1) / BN;
  p.tiles_n = (int)tiles_n;
  p.full_tiles = (int)(tiles_m * tiles_n);
  p.tail_s = 1;
  p.tile_bm = BM; p.tile_bn = BN;
  long grid = tiles_m * tiles_n;
  if (grid > 256) grid = 256;                   // one persistent workgroup per CU (a multiple of 8: the XCD chunking relies on it)
  const int epi = d->geglu ? 2 : (d->act == TG_ACT_NONE ? 0 : 1);
  if (epi == 0) launch_bt_kernel<T, BM, BN, WM, WN, 0>(p, grid, lds, st);
  else if (epi == 2) {
    if constexpr ((BN / (WN * 32)) % 2 == 0) launch_bt_kernel<T, BM, BN, WM, WN, 2>(p, grid, lds, st);
    else { tg_set_error("tg_gemm: this big tile has no GEGLU epilogue"); return TG_ERR_UNSUPPORTED; }
  } else launch_bt_kernel<T, BM, BN, WM, WN, 1>(p, grid, lds, st);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

}  // namespace

// Called by tg_gemm.hip's planner (not part of the C ABI): tile 1 = 128 x 320, 2 = 256 x 256 (GEGLU-capable).  (A 256 x 320
// tile — 160 accumulator registers per lane — was built and dropped: it spills inside the K loop at the 256-register budget
// of two waves per SIMD, and a spill reload is a VMEM access whose wait drains the in-flight LDS-DMA.)
// GemmParams arrives filled except for the tile bookkeeping.
int tg_gemm_bt_launch(const tg_gemm_desc* d, const void* params, int bt_tile, void* stream) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (bt_tile != 2) { tg_set_error("tg_gemm: unknown big tile %d (the 128 x 320 instance was removed in round 5: never selected)", bt_tile); return TG_ERR_ARG; }
  if (d->dtype == TG_BF16) {
    return launch_bt<bf16_t, 256, 256, 4, 2>(d, p, st);
  }
  return launch_bt<f16_t, 256, 256, 4, 2>(d, p, st);
}

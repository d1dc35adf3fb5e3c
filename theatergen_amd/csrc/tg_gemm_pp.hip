// "Ping-pong" MFMA GEMM for gfx950 (round 6; include/theatergen_hip.h: tg_gemm, selected by the planner in tg_gemm.hip — pp_variant_of).
//
// Why a third plain-GEMM kernel.  The LDS-DMA kernels of tg_gemm_glds.h (128 x 128 / 128 x 160, four waves) and the lock-step big tile of
// tg_gemm_bt.hip top out at 0.85 - 1.0 PFLOP/s on random bf16 although the matrix pipe alone sustains 1.88 PFLOP/s on the same data and the
// vendor's bare matmul reaches 1.5 - 1.6 (profiles/r6_mfma_sustained.json): in all of them the two (or one) waves of a SIMD read fragments at the
// same time and issue MFMAs at the same time, so the pipe idles through every fragment-read / DMA-issue / barrier stretch.  Here ONE 8-wave
// workgroup per CU owns a 256 x 256 output tile and its two halves — waves 0-3 and waves 4-7, one wave of each per SIMD — run the SAME phase
// program ONE BARRIER APART (cdna_hip_programming.md section 5, "8-phase" schedule): while a wave issues the 16 MFMAs of a C quadrant its SIMD
// partner reads the next quadrant's fragments and issues its share of the operand LDS-DMA, then they swap.  First build, isolated, random
// operands (scripts/dev_gemm8.py): 8192 x 4096 x 4096 1298 TFLOP/s against 847 (128 x 128) and 1518 (vendor); 4096 x 3840 x 1280 42 us against 55.
//
//   * tile 256 x 256 x 64, waves 2 (M) x 4 (N), wave tile 128 x 64 = 8 x 4 MFMA tiles of 16 x 16 x 32 (128 accumulator registers), issued "swapped"
//     (A operand = W rows) like every other GEMM here: lane & 15 = token, 4 consecutive registers = 4 consecutive output channels;
//   * four phases per K-tile = the four 64 x 32 C quadrants in Gray order (0,0) (0,1) (1,1) (1,0): a phase reads 12 / 4 / 8 / 4 fragments
//     (ds_read_b128, 128-byte rows XOR-swizzled on the DMA source address and on the read), waits for them, crosses a barrier, issues 16 MFMAs,
//     crosses a barrier;
//   * operands HBM / L2 -> LDS by global_load_lds_dwordx4 into half-tile slots (A rows 0-127 / 128-255, W rows 0-127 / 128-255) of two K-tile
//     parities (128 KB): the A halves of K-tile t + 2 are requested in phase 4 of tile t (their slots were read for the last time in phase 3), the
//     W halves in phase 2 of tile t + 1; ONE counted vmcnt per K-tile (phase 4, before its first barrier) leaves the youngest request in flight;
//   * persistent, XCD-chunked tile walk (column-major when the weight is the larger operand: an XCD then owns a weight column panel that fits its
//     L2 and the activation rows stream); K-tile 0 of the next output tile is requested before the epilogue, K-tile 1 behind it;
//   * epilogue through an LDS bounce (32 tokens x 64 channels per wave and pass, in the dead parity-1 slots + the 32 KB above the stages): bias /
//     per-batch vector / residual loads and the stores are 16 bytes per lane on whole 128-byte rows; GEGLU (models/attention.py:337-338; weight rows
//     packed [a(32) ; gate(32)] per 64) pairs quadrant (q, 0) with (q, 1) in the accumulator layout before the bounce; fp32 throughout, one rounding.
#include "tg_gemm_common.h"

typedef __attribute__((ext_vector_type(2))) float f32x2;

namespace {

template <typename T, int EPI, int LN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void pp_gemm_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 256;
  constexpr unsigned HALF = 128 * 128;            // bytes of a half-tile (128 rows x 64 k)
  constexpr unsigned PARITY = 4 * HALF;           // one K-tile: A half 0, A half 1, W half 0, W half 1
  constexpr int NH = 2;                           // LDS-DMA instructions per wave per half-tile (16 pieces of 1 KiB over 8 waves)
  typedef typename Vec<T>::v8 V8;
  extern __shared__ __attribute__((aligned(128))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int frow = lane & 15, fq = lane >> 4;
  const int tiles_m = (int)(p.M / BM), tiles_n = p.tiles_n;
  const int ntiles = tiles_m * tiles_n;
  const int nkt = (int)(p.K / BK);
  const T* A0 = reinterpret_cast<const T*>(p.a0);
  const T* Wp = reinterpret_cast<const T*>(p.w);

  auto tile_of = [&](int v, long& m0, long& n0) {
    const int lb = xcd_chunked_block_id(v, ntiles);
    int tm, tn;
    if (p.slab_order == 1) { tn = lb / tiles_m; tm = lb - tn * tiles_m; }      // column-major: an XCD's chunk is a band of weight columns
    else { tm = lb / tiles_n; tn = lb - tm * tiles_n; }
    m0 = (long)tm * BM; n0 = (long)tn * BN;
  };

  // ---- LDS-DMA: piece q (1 KiB) = rows [8q, 8q + 8) of a half-tile; lane -> (row 8q + lane / 8, slot lane % 8); the 16-byte chunk fetched into a
  // slot is slot ^ key(row), key(row) = (row >> 1) & 7 (inline asm: hipcc's waitcnt pass must not see these, tg_gemm_bt.hip)
  // the source of a request = a wave-uniform base (SGPR pair: operand pointer + the K-tile's byte offset) + this lane's 32-bit byte offset (the
  // row of its piece and its swizzled 16-byte chunk): eight offset registers per lane for the eight requests of a K-tile, no address VALU in the loop
  auto dma = [&](const T* base, unsigned voff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_byte_addr) : "memory");
  };
  unsigned soff[4][NH];                           // byte offset of this lane's 16 bytes of K-tile 0 for its pieces of the four half-tiles
  auto setup_tile = [&](long m0, long n0) {
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
      for (int j = 0; j < NH; ++j) {
        const int r = (j * 8 + wave) * 8 + (lane >> 3);                      // row inside the half
        const int ch = (lane & 7) ^ ((r >> 1) & 7);
        if (h < 2) {
          const long m = m0 + h * 128 + r;
          long off = m * p.lda;
          if (p.a_rpb > 0) { const long bb = m / p.a_rpb; off = bb * p.a_bs + (m - bb * p.a_rpb) * p.lda; }
          soff[h][j] = (unsigned)((off + ch * 8) * (long)sizeof(T));
        } else {
          soff[h][j] = (unsigned)(((n0 + (h - 2) * 128 + r) * p.ldw + ch * 8) * (long)sizeof(T));
        }
      }
  };
  auto issue_half = [&](int h, int kt) {
    const unsigned dst = lds0 + (unsigned)(kt & 1) * PARITY + (unsigned)h * HALF + (unsigned)wave * 1024u;
    const T* base = (h < 2 ? A0 : Wp) + (long)kt * BK;
#pragma unroll
    for (int j = 0; j < NH; ++j) dma(base, soff[h][j], dst + (unsigned)j * 8192u);
  };

  // ---- fragments: lane -> row (lane & 15) of a 16-row MFMA tile, 16-byte chunk 4 ks + (lane >> 4) of its 128-byte row.  Row r of a half sits at
  // r * 128, its chunk c in slot c ^ key(r); key depends on bits 1..3 of the row only, so the tiles of a quadrant are immediate offsets apart and
  // the second k-step is the first with byte-address bit 6 flipped.
  const unsigned fsw = (unsigned)((fq ^ ((frow >> 1) & 7)) << 4);
  const unsigned a_base = lds0 + (unsigned)wr * HALF + (unsigned)frow * 128u + fsw;
  const unsigned b_base = lds0 + 2u * HALF + (unsigned)(wc >> 1) * HALF + (unsigned)((wc & 1) * 64 + frow) * 128u + fsw;
  u32x4 af[4][2], bf[2][2];
#define PP_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  auto read_a = [&](unsigned par, int qa) {       // rows qa * 64 + i * 16 + frow
    const unsigned a0 = a_base + par + (unsigned)qa * 8192u, a1 = a0 ^ 64u;
    PP_READ(af[0][0], a0, 0); PP_READ(af[0][1], a1, 0);
    PP_READ(af[1][0], a0, 2048); PP_READ(af[1][1], a1, 2048);
    PP_READ(af[2][0], a0, 4096); PP_READ(af[2][1], a1, 4096);
    PP_READ(af[3][0], a0, 6144); PP_READ(af[3][1], a1, 6144);
  };
  auto read_b = [&](unsigned par, int qb) {       // W rows (wc & 1) * 64 + qb * 32 + j * 16 + frow
    const unsigned b0 = b_base + par + (unsigned)qb * 4096u, b1 = b0 ^ 64u;
    PP_READ(bf[0][0], b0, 0); PP_READ(bf[0][1], b1, 0);
    PP_READ(bf[1][0], b0, 2048); PP_READ(bf[1][1], b1, 2048);
  };
#undef PP_READ
  f32x4 acc[2][2][4][2];                          // [qa][qb][i][j]: tokens qa * 64 + i * 16 + frow, channels qb * 32 + j * 16 + 4 fq + e
  // s_setprio 1 around a phase's MFMA block (guide T5: the phase split gives the arbiter something to prefer): same-box A/B 842.8 -> 839.9 ms per story
  // (profiles/r6_ab_prio.json); TG_GEMM_FLAGS bit 15 (dev) switches it off
  const bool prio = (p.flags & 32768) == 0;
  auto mfma_quad = [&](int qa, int qb) {
    if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[qa][qb][i][j] = mfma16(__builtin_bit_cast(V8, bf[j][ks]), __builtin_bit_cast(V8, af[i][ks]), acc[qa][qb][i][j]);
    if (prio) __builtin_amdgcn_s_setprio(0);
  };
  // end of a phase's load section: my fragment reads have RETURNED (so a slot may be refilled one phase after its last read, and the MFMAs below
  // may use them), then the workgroup barrier; the scheduling fences keep hipcc from moving MFMAs / reads across (guide 5.4 rule 18)
#define PP_LOAD_END()                                      \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();                          \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)
#define PP_MFMA_END()                                      \
  do {                                                     \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();                          \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)

  int v = blockIdx.x;
  long m0, n0;
  tile_of(v, m0, n0);
  setup_tile(m0, n0);
  issue_half(0, 0); issue_half(1, 0); issue_half(2, 0); issue_half(3, 0);
  if (nkt > 1) { issue_half(0, 1); issue_half(1, 1); issue_half(2, 1); issue_half(3, 1); }
  for (; v < ntiles; v += gridDim.x) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // K-tile 0 has landed (K-tile 1, the younger 4 NH requests, may stay in flight)
    if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NH) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();    // the second group runs one barrier behind the first
    __builtin_amdgcn_sched_barrier(0);
    for (int t = 0; t < nkt; ++t) {
      const unsigned par = (unsigned)(t & 1) * PARITY;
      // phase 1: quadrant (0, 0)
      read_b(par, 0);
      read_a(par, 0);
      PP_LOAD_END();
      mfma_quad(0, 0);
      PP_MFMA_END();
      // phase 2: quadrant (0, 1); the W halves of K-tile t + 1 (their slots were read for the last time in phase 4 of tile t - 1)
      read_b(par, 1);
      if (t >= 1 && t + 1 < nkt) { issue_half(2, t + 1); issue_half(3, t + 1); }
      PP_LOAD_END();
      mfma_quad(0, 1);
      PP_MFMA_END();
      // phase 3: quadrant (1, 1)
      read_a(par, 1);
      PP_LOAD_END();
      mfma_quad(1, 1);
      PP_MFMA_END();
      // phase 4: quadrant (1, 0); the A halves of K-tile t + 2 (this parity's A slots were read for the last time in phase 3); K-tile t + 1 landed
      read_b(par, 0);
      if (t + 2 < nkt) {
        issue_half(0, t + 2); issue_half(1, t + 2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NH) : "memory");
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PP_LOAD_END();
      mfma_quad(1, 0);
      PP_MFMA_END();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();    // pairs with the second group's last barrier: every fragment read of the tile is done
    __builtin_amdgcn_sched_barrier(0);

    // ---- next output tile: K-tile 0 into parity 0 under the epilogue (which bounces through parity 1 and the 32 KB above the stages)
    const long pm0 = m0, pn0 = n0;
    const int vn = v + (int)gridDim.x;
    if (vn < ntiles) {
      tile_of(vn, m0, n0);
      setup_tile(m0, n0);
      issue_half(0, 0); issue_half(1, 0); issue_half(2, 0); issue_half(3, 0);
    }

    // ---- epilogue
    float* scr = reinterpret_cast<float*>(smem + PARITY) + wave * 3072;     // 12 KiB per wave
    const long m_w = pm0 + wr * 128, n_w = pn0 + wc * 64;
    T* outp = reinterpret_cast<T*>(p.out);
    const T* biasp = reinterpret_cast<const T*>(p.bias);
    const float scale = p.out_scale;
    if constexpr (LN == 2) {
      // LayerNorm fold with precomputed row statistics, in the accumulator layout: LN(x) W^T = rstd (x W'^T) + (-rstd mean) u + v
      // (W' = W gamma, u = row sums of W', v = W beta; tg_gemm_glds.h).  Per lane: the (rstd, -rstd mean) pairs of its 8 token rows, then one
      // 16-column group of u / v at a time; v rides with the bias below.
#pragma unroll
      for (int qa = 0; qa < 2; ++qa) {
        float rs[4], rm[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x2 st = *reinterpret_cast<const f32x2*>(p.ln_rows + 2 * (m_w + qa * 64 + i * 16 + frow));
          rs[i] = st[0]; rm[i] = st[1];
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const long col = n_w + qb * 32 + j * 16 + 4 * fq;
            const f32x4 u4 = *reinterpret_cast<const f32x4*>(p.ln_u + col), v4 = *reinterpret_cast<const f32x4*>(p.ln_v + col);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[qa][qb][i][j][e] = rs[i] * acc[qa][qb][i][j][e] + (rm[i] * u4[e] + v4[e]);
          }
      }
      __builtin_amdgcn_sched_barrier(0);          // the epilogue's own loads are not hoisted over the fold (they would sit next to all 128 accumulators)
    }
    if constexpr (EPI == 2) {
      // GEGLU in the accumulator layout: a = quadrant (q, 0), gate = quadrant (q, 1), same lane / register -> 32 output channels per wave
      float ba[2][4], bg[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ba[j][e] = biasp ? to_f32<T>(biasp[n_w + j * 16 + 4 * fq + e]) : 0.f;
          bg[j][e] = biasp ? to_f32<T>(biasp[n_w + 32 + j * 16 + 4 * fq + e]) : 0.f;
        }
      constexpr int RS = 36;
      const int c = lane & 3, r0 = lane >> 2;     // 4 pieces of 8 channels x 16 rows per pass
#pragma unroll
      for (int b = 0; b < 4; ++b) {               // 32-token blocks of the wave tile
        const int qa = b >> 1;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int i = 2 * (b & 1) + ii;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float a = acc[qa][0][i][j][e], g = acc[qa][1][i][j][e];
              o[e] = (a + ba[j][e]) * gelu_erf_f(g + bg[j][e]) * scale;
            }
            *reinterpret_cast<f32x4*>(scr + (16 * ii + frow) * RS + 16 * j + 4 * fq) = o;
          }
        }
        __builtin_amdgcn_wave_barrier();
        f32x4 lo[2], hi[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          lo[it] = *reinterpret_cast<const f32x4*>(scr + (it * 16 + r0) * RS + c * 8);
          hi[it] = *reinterpret_cast<const f32x4*>(scr + (it * 16 + r0) * RS + c * 8 + 4);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const long m = m_w + 32 * b + it * 16 + r0;
          V8 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) { o[e] = from_f32<T>(lo[it][e]); o[4 + e] = from_f32<T>(hi[it][e]); }
          *reinterpret_cast<V8*>(outp + m * p.ldc + (n_w >> 1) + c * 8) = o;
        }
      }
    } else {
      constexpr int RS = 68;
      const int c = lane & 7, r0 = lane >> 3;     // 8 pieces of 8 channels x 8 rows per pass
      const long n = n_w + c * 8;
      const T* bvecp = reinterpret_cast<const T*>(p.bvec);
      const T* resp = reinterpret_cast<const T*>(p.res);
      float bias_f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bias_f[e] = 0.f;
      if (biasp != nullptr) {
        const V8 b8 = *reinterpret_cast<const V8*>(biasp + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) bias_f[e] = to_f32<T>(b8[e]);
      }
      const bool to_t = p.n_split > 0 && n_w >= p.n_split;                    // this wave's 64 columns go to the transposed output (V^T)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int qa = b >> 1;
        const long mb = m_w + 32 * b;
        if (to_t) {
          // lane = token is already the contiguous direction of out_t[(batch, column), token]: direct 2-byte stores, 16 tokens per run
          // (32-bit element offsets from out_t: one VGPR add per store instead of a 64-bit address each — the planner checks the extent)
          T* ot = reinterpret_cast<T*>(p.out_t);
          const unsigned nt = (unsigned)(p.N - p.n_split), ldt = (unsigned)p.ldt;
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * (b & 1) + ii;
            const long m = mb + 16 * ii + frow;
            const unsigned bb = (unsigned)(m / p.rows_per_batch), tok = (unsigned)(m - (long)bb * p.rows_per_batch);
            const unsigned col0 = (unsigned)(n_w - p.n_split) + 4u * (unsigned)fq;
            const unsigned o0 = (bb * nt + col0) * ldt + tok;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                float bq[4] = {0.f, 0.f, 0.f, 0.f};
                if (biasp != nullptr) {
                  const typename Vec<T>::v4 b4 = *reinterpret_cast<const typename Vec<T>::v4*>(biasp + n_w + qb * 32 + j * 16 + 4 * fq);
#pragma unroll
                  for (int e = 0; e < 4; ++e) bq[e] = to_f32<T>(b4[e]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  ot[o0 + (unsigned)(qb * 32 + j * 16 + e) * ldt] = from_f32<T>((acc[qa][qb][i][j][e] + bq[e]) * scale);
              }
          }
          continue;
        }
        V8 res8[4];
        if (resp != nullptr) {
#pragma unroll
          for (int it = 0; it < 4; ++it) res8[it] = *reinterpret_cast<const V8*>(resp + (mb + it * 8 + r0) * p.ldres + n);
        }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              *reinterpret_cast<f32x4*>(scr + (16 * ii + frow) * RS + qb * 32 + j * 16 + 4 * fq) = acc[qa][qb][2 * (b & 1) + ii][j];
        __builtin_amdgcn_wave_barrier();
        // one pass = 8 rows x 8 pieces: read, finish, store (the reads of a pass are not hoisted over the previous pass's arithmetic: the block would
        // hold 32 more registers next to the 128 accumulators and the kernel would spill inside its K loop)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const f32x4 lo_ = *reinterpret_cast<const f32x4*>(scr + (it * 8 + r0) * RS + c * 8);
          const f32x4 hi_ = *reinterpret_cast<const f32x4*>(scr + (it * 8 + r0) * RS + c * 8 + 4);
          float x[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { x[e] = lo_[e]; x[4 + e] = hi_[e]; }
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] += bias_f[e];
          if (bvecp != nullptr) {       // per-batch vector (rare on plain GEMMs): loaded per pass, not held next to the accumulators
            const V8 a8 = *reinterpret_cast<const V8*>(bvecp + ((mb + it * 8 + r0) / p.rows_per_batch) * p.ldbvec + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] += to_f32<T>(a8[e]);
          }
          if (resp != nullptr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] += to_f32<T>(res8[it][e]);
          }
          if constexpr (EPI == 1) {
            if (p.act != TG_ACT_NONE) {
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] = apply_act(x[e], p.act);
            }
          }
          V8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(x[e] * scale);
          *reinterpret_cast<V8*>(outp + (mb + it * 8 + r0) * p.ldc + n) = o;
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_wave_barrier();          // the next block overwrites the bounce
      }
    }
    // every wave is out of its bounce before K-tile 1 of the next output tile lands in parity 1; the epilogue's own loads / stores are drained so
    // that the counted waits of the next tile see LDS-DMA requests only
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (vn < ntiles && nkt > 1) { issue_half(0, 1); issue_half(1, 1); issue_half(2, 1); issue_half(3, 1); }
  }
#undef PP_LOAD_END
#undef PP_MFMA_END
}

template <typename T, int EPI, int LN>
int launch_pp(const GemmParams& p0, hipStream_t st) {
  GemmParams p = p0;
  constexpr size_t lds = 160 * 1024;
  const long tiles_m = p.M / 256, tiles_n = p.N / 256;
  p.tiles_n = (int)tiles_n;
  p.tile_bm = 256; p.tile_bn = 256; p.full_tiles = (int)(tiles_m * tiles_n); p.tail_s = 1;
  // tile walk: column-major when the weight is the larger operand (an XCD's chunk then covers whole weight column panels: FeedForward net.0 of the
  // 16 x 16 level, 4096 x 10240 x 1280, moved 578 MB per launch against 79 MB algorithmic on row-major 128 x 128 tiles, profiles/r5_pmc_traffic.json)
  p.slab_order = ((long)p.N > p.M && !(p.flags & 8192)) ? 1 : 0;
  long grid = tiles_m * tiles_n;
  if (grid > 256) grid = 256;
  auto k = pp_gemm_kernel<T, EPI, LN>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr;
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(512), lds, st, p);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

template <typename T>
int launch_pp_dtype(const tg_gemm_desc* d, const GemmParams& p, hipStream_t st) {
  const bool ln = d->ln_u != nullptr;
  if (d->geglu) return ln ? launch_pp<T, 2, 2>(p, st) : launch_pp<T, 2, 0>(p, st);
  if (d->act != TG_ACT_NONE) return launch_pp<T, 1, 0>(p, st);
  return ln ? launch_pp<T, 0, 2>(p, st) : launch_pp<T, 0, 0>(p, st);
}

}  // namespace

// Called by tg_gemm.hip's planner (not part of the C ABI); GemmParams arrives filled except for the tile bookkeeping.  Takes plain single-source GEMMs
// with M % 256 == 0, N % 256 == 0, K % 64 == 0, 16-byte aligned operands / pitches (p.epi_lds), linear / activation / GEGLU epilogues, the
// transposed V^T columns (n_split % 64 == 0) and the LayerNorm fold with PRECOMPUTED row statistics (ln_rows).
int tg_gemm_pp_launch(const tg_gemm_desc* d, const void* params, void* stream) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == TG_BF16) return launch_pp_dtype<bf16_t>(d, p, st);
  return launch_pp_dtype<f16_t>(d, p, st);
}

// Ping-pong MFMA GEMM on 256 x 160 tiles (round 6; planner: tg_gemm.hip pp160_selected, force_tile 25) — tg_gemm_pp.hip's structure for the UNet's
// N = 640 / 1920 projections, whose tile COUNT (round 5, tg_gemm_t160.hip) only works out on 160-column tiles: 16384 x 640 is 256 tiles of 256 x 160 =
// exactly one per CU, 16384 x 1920 is 768 = three rounds.
//   * tile 256 x 160 x 64, 8 waves = 4 (M) x 2 (N), wave tile 64 x 80 = 4 x 5 MFMA tiles of 16 x 16 x 32 (80 accumulator registers), issued "swapped":
//     lane & 15 = token, 4 consecutive registers = 4 consecutive output channels; the two waves of a SIMD hold the two column halves of the same rows;
//   * two phases per K-tile = its two 32-deep k-steps: a phase reads 4 activation + 5 weight fragments, waits for them, crosses a barrier, issues 20
//     MFMAs, crosses a barrier; waves 4-7 run the same program one barrier behind waves 0-3 (one wave's MFMAs cover its SIMD partner's reads);
//   * operands HBM / L2 -> LDS by global_load_lds_dwordx4 into a ring of THREE 52 KB K-tile stages (156 KB): K-tile t + 2 is requested during tile t (its
//     stage was read for the last time during tile t - 1: no intra-tile write-after-read schedule), ONE counted vmcnt per K-tile; 52 one-KiB pieces per
//     K-tile over 8 waves = 7 requests per wave (pieces 52 .. 55 re-request pieces 48 .. 51: same bytes to the same place);
//   * one output tile per workgroup (the shapes this kernel is selected for are one to three rounds), XCD-chunked tile order;
//   * epilogue: fp32 bounce through the dead stages, 16 tokens x 64 channels per wave and pass + one 32 x 16 pass for the last 16 channels of the wave's
//     80: bias / per-batch vector / residual loads and the stores are 16 bytes per lane; V^T columns (n_split on a multiple of 80) leave through direct
//     2-byte stores (lane = token is their contiguous direction); the LayerNorm fold from precomputed row statistics in the accumulator layout.
#include "tg_gemm_common.h"

typedef __attribute__((ext_vector_type(2))) float f32x2_t;

namespace {

template <typename T, int EPI, int LN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void pp160_gemm_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 160;
  constexpr unsigned A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;       // 32 KB + 20 KB
  constexpr int NP = 7;                           // LDS-DMA requests per wave per K-tile (52 pieces over 8 waves, the last four doubled)
  typedef typename Vec<T>::v8 V8;
  typedef typename Vec<T>::v4 V4;
  extern __shared__ __attribute__((aligned(128))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int group = wave >> 2;                    // = column half of the tile (waves w and w + 4 share a SIMD)
  const int wm = wave & 3;                        // 64-row block of the tile
  const int frow = lane & 15, fq = lane >> 4;
  const int tiles_m = (int)(p.M / BM), tiles_n = p.tiles_n;
  const int ntiles = tiles_m * tiles_n;
  const int nkt = (int)(p.K / BK);
  const T* A0 = reinterpret_cast<const T*>(p.a0);
  const T* Wp = reinterpret_cast<const T*>(p.w);

  const int lb = xcd_chunked_block_id(blockIdx.x, ntiles);
  int tm, tn;
  if (p.slab_order == 1) { tn = lb / tiles_m; tm = lb - tn * tiles_m; }
  else { tm = lb / tiles_n; tn = lb - tm * tiles_n; }
  const long m0 = (long)tm * BM, n0 = (long)tn * BN;

  // ---- LDS-DMA: piece q (1 KiB) = rows [8q, 8q + 8) of the K-tile image (rows 0-255 activations, 256-415 weights); lane -> (row 8q + lane / 8, slot
  // lane % 8), the 16-byte chunk fetched into a slot is slot ^ key(row), key(row) = (row >> 1) & 7.  Source = SGPR base (operand + K-tile offset) + this
  // lane's 32-bit byte offset (tg_gemm_pp.hip).
  auto dma = [&](const T* base, unsigned voff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_byte_addr) : "memory");
  };
  unsigned soff[NP];                              // request j of this wave = piece q = 8 j + wave (q >= 52: piece q - 4 again)
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    int q = 8 * j + wave;
    if (q >= 52) q -= 4;
    const int r = q * 8 + (lane >> 3);            // row of the K-tile image
    const int ch = (lane & 7) ^ ((r >> 1) & 7);
    if (q < 32) {
      const long m = m0 + r;
      long off = m * p.lda;
      if (p.a_rpb > 0) { const long bb = m / p.a_rpb; off = bb * p.a_bs + (m - bb * p.a_rpb) * p.lda; }
      soff[j] = (unsigned)((off + ch * 8) * (long)sizeof(T));
    } else {
      soff[j] = (unsigned)(((n0 + (r - 256)) * p.ldw + ch * 8) * (long)sizeof(T));
    }
  }
  auto issue = [&](int kt, int j0, int j1) {      // requests [j0, j1) of K-tile kt into stage kt % 3
    const unsigned st = lds0 + (unsigned)(kt % 3) * STAGE;
    const T* ab = A0 + (long)kt * BK;
    const T* wb = Wp + (long)kt * BK;
#pragma unroll
    for (int j = j0; j < j1; ++j) {
      int q = 8 * j + wave;                        // (wave-uniform)
      if (q >= 52) q -= 4;
      dma(q < 32 ? ab : wb, soff[j], st + (unsigned)q * 1024u);
    }
  };

  // ---- fragments: lane -> row (lane & 15) of a 16-row MFMA tile, chunk 4 ks + (lane >> 4) in slot chunk ^ key(row); key from bits 1..3 of the row only
  const unsigned fsw = (unsigned)((fq ^ ((frow >> 1) & 7)) << 4);
  const unsigned a_base = lds0 + (unsigned)(wm * 64 + frow) * 128u + fsw;
  const unsigned b_base = lds0 + A_BYTES + (unsigned)(group * 80 + frow) * 128u + fsw;
  u32x4 xf[4], wf[5];
  f32x4 acc[4][5];
#define P160_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  auto read_frags = [&](unsigned st, int ks) {
    const unsigned a = (a_base + st) ^ ((unsigned)ks << 6), b = (b_base + st) ^ ((unsigned)ks << 6);
    P160_READ(wf[0], b, 0); P160_READ(wf[1], b, 2048); P160_READ(wf[2], b, 4096); P160_READ(wf[3], b, 6144); P160_READ(wf[4], b, 8192);
    P160_READ(xf[0], a, 0); P160_READ(xf[1], a, 2048); P160_READ(xf[2], a, 4096); P160_READ(xf[3], a, 6144);
  };
#undef P160_READ
  const bool prio = (p.flags & 32768) == 0;       // s_setprio 1 around a phase's MFMA block (tg_gemm_pp.hip); TG_GEMM_FLAGS bit 15 (dev) switches it off
  auto mfmas = [&]() {
    if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[i][j] = mfma16(__builtin_bit_cast(V8, wf[j]), __builtin_bit_cast(V8, xf[i]), acc[i][j]);
    if (prio) __builtin_amdgcn_s_setprio(0);
  };
#define P160_LOAD_END()                                    \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();                          \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)
#define P160_MFMA_END()                                    \
  do {                                                     \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();                          \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  issue(0, 0, NP);
  if (nkt > 1) issue(1, 0, NP);
  if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (group == 1) __builtin_amdgcn_s_barrier();   // the second group runs one barrier behind the first
  __builtin_amdgcn_sched_barrier(0);
  for (int t = 0; t < nkt; ++t) {
    const unsigned st = (unsigned)(t % 3) * STAGE;
    // phase 1: k-step 0; first part of K-tile t + 2 (its stage was read for the last time during tile t - 1)
    read_frags(st, 0);
    if (t + 2 < nkt) issue(t + 2, 0, 4);
    P160_LOAD_END();
    mfmas();
    P160_MFMA_END();
    // phase 2: k-step 1; the rest of K-tile t + 2; K-tile t + 1 (requested during tile t - 1) has landed before this phase's first barrier
    read_frags(st, 1);
    if (t + 2 < nkt) {
      issue(t + 2, 4, NP);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P160_LOAD_END();
    mfmas();
    P160_MFMA_END();
  }
  if (group == 0) __builtin_amdgcn_s_barrier();   // pairs with the second group's last barrier: every fragment read is done, the stages are dead
  __builtin_amdgcn_sched_barrier(0);
#undef P160_LOAD_END
#undef P160_MFMA_END

  // ---- epilogue
  float* scr = reinterpret_cast<float*>(smem) + wave * (16 * 68);
  const long m_w = m0 + wm * 64, n_w = n0 + group * 80;
  T* outp = reinterpret_cast<T*>(p.out);
  const T* biasp = reinterpret_cast<const T*>(p.bias);
  const T* bvecp = reinterpret_cast<const T*>(p.bvec);
  const T* resp = reinterpret_cast<const T*>(p.res);
  const float scale = p.out_scale;
  if constexpr (LN == 2) {
    // LayerNorm fold with precomputed row statistics, in the accumulator layout: LN(x) W^T = rstd (x W'^T) + (-rstd mean) u + v (tg_gemm_pp.hip)
    float rs[4], rm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x2_t s2 = *reinterpret_cast<const f32x2_t*>(p.ln_rows + 2 * (m_w + i * 16 + frow));
      rs[i] = s2[0]; rm[i] = s2[1];
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const long col = n_w + j * 16 + 4 * fq;
      const f32x4 u4 = *reinterpret_cast<const f32x4*>(p.ln_u + col), v4 = *reinterpret_cast<const f32x4*>(p.ln_v + col);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = rs[i] * acc[i][j][e] + (rm[i] * u4[e] + v4[e]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (p.n_split > 0 && n_w >= p.n_split) {
    // this wave's 80 columns go to the transposed output (V^T): lane = token is its contiguous direction; 32-bit element offsets from out_t
    T* ot = reinterpret_cast<T*>(p.out_t);
    const unsigned nt = (unsigned)(p.N - p.n_split), ldt = (unsigned)p.ldt;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long m = m_w + 16 * i + frow;
      const unsigned bb = (unsigned)(m / p.rows_per_batch), tok = (unsigned)(m - (long)bb * p.rows_per_batch);
      const unsigned o0 = (bb * nt + (unsigned)(n_w - p.n_split) + 4u * (unsigned)fq) * ldt + tok;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        float bq[4] = {0.f, 0.f, 0.f, 0.f};
        if (biasp != nullptr) {
          const V4 b4 = *reinterpret_cast<const V4*>(biasp + n_w + j * 16 + 4 * fq);
#pragma unroll
          for (int e = 0; e < 4; ++e) bq[e] = to_f32<T>(b4[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) ot[o0 + (unsigned)(j * 16 + e) * ldt] = from_f32<T>((acc[i][j][e] + bq[e]) * scale);
      }
    }
    return;
  }
  auto finish8 = [&](const f32x4& lo, const f32x4& hi, long m, long n, const float (&bias_f)[8]) {
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
    if constexpr (EPI == 1) {
      if (p.act != TG_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = apply_act(x[e], p.act);
      }
    }
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(x[e] * scale);
    *reinterpret_cast<V8*>(outp + m * p.ldc + n) = o;
  };
  auto bias8 = [&](long n, float (&bias_f)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) bias_f[e] = 0.f;
    if (biasp != nullptr) {
      const V8 b8 = *reinterpret_cast<const V8*>(biasp + n);
#pragma unroll
      for (int e = 0; e < 8; ++e) bias_f[e] = to_f32<T>(b8[e]);
    }
  };
  {
    // channels 0-63 of the wave's 80: 16 tokens x 64 channels per pass, lane -> (row lane / 8 + 8 it, 8-channel piece lane % 8)
    const int c = lane & 7, r0 = lane >> 3;
    const long n = n_w + c * 8;
    float bias_f[8];
    bias8(n, bias_f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(scr + frow * 68 + 16 * j + 4 * fq) = acc[i][j];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(scr + (it * 8 + r0) * 68 + c * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(scr + (it * 8 + r0) * 68 + c * 8 + 4);
        finish8(lo, hi, m_w + i * 16 + it * 8 + r0, n, bias_f);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  {
    // channels 64-79: two row tiles per pass = 32 tokens x 16 channels, lane -> (row lane / 2, 8-channel piece lane % 2)
    const int c = lane & 1, r = lane >> 1;
    const long n = n_w + 64 + c * 8;
    float bias_f[8];
    bias8(n, bias_f);
#pragma unroll
    for (int ip = 0; ip < 2; ++ip) {
      *reinterpret_cast<f32x4*>(scr + frow * 20 + 4 * fq) = acc[2 * ip][4];
      *reinterpret_cast<f32x4*>(scr + (16 + frow) * 20 + 4 * fq) = acc[2 * ip + 1][4];
      __builtin_amdgcn_wave_barrier();
      const f32x4 lo = *reinterpret_cast<const f32x4*>(scr + r * 20 + c * 8);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(scr + r * 20 + c * 8 + 4);
      finish8(lo, hi, m_w + ip * 32 + r, n, bias_f);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <typename T, int EPI, int LN>
int launch_pp160(const GemmParams& p0, hipStream_t st) {
  GemmParams p = p0;
  constexpr size_t lds = 3 * (256 + 160) * 128;
  const long tiles_m = p.M / 256, tiles_n = p.N / 160;
  p.tiles_n = (int)tiles_n;
  p.tile_bm = 256; p.tile_bn = 160; p.full_tiles = (int)(tiles_m * tiles_n); p.tail_s = 1;
  p.slab_order = ((long)p.N > p.M && !(p.flags & 8192)) ? 1 : 0;
  auto k = pp160_gemm_kernel<T, EPI, LN>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr;
  hipLaunchKernelGGL(k, dim3((unsigned)(tiles_m * tiles_n)), dim3(512), lds, st, p);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

template <typename T>
int launch_pp160_dtype(const tg_gemm_desc* d, const GemmParams& p, hipStream_t st) {
  const bool ln = d->ln_u != nullptr;
  if (d->act != TG_ACT_NONE) return launch_pp160<T, 1, 0>(p, st);
  return ln ? launch_pp160<T, 0, 2>(p, st) : launch_pp160<T, 0, 0>(p, st);
}

}  // namespace

// Called by tg_gemm.hip's planner (not part of the C ABI).  Takes plain single-source GEMMs with M % 256 == 0, N % 160 == 0, K % 64 == 0, 16-byte aligned
// operands / pitches, linear / activation epilogues, V^T columns (n_split % 80 == 0) and the LayerNorm fold with precomputed row statistics (no GEGLU).
int tg_gemm_pp160_launch(const tg_gemm_desc* d, const void* params, void* stream) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == TG_BF16) return launch_pp160_dtype<bf16_t>(d, p, st);
  return launch_pp160_dtype<f16_t>(d, p, st);
}

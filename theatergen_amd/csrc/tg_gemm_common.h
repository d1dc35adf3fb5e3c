// Shared pieces of the MFMA GEMM / conv kernels (tg_gemm.hip, tg_gemm_bt.hip): launch parameters, the XCD-aware tile order,
// the epilogues (direct and LDS-transposed), the zero page of the LDS-DMA gathers.  Everything lives in an anonymous
// namespace: each translation unit gets its own copy (the kernels of the two files share no device symbols).
#pragma once
#include "tg_common.h"

namespace {

constexpr int BK = 64;

struct GemmParams {
  const void* a0;
  const void* a1;
  int c0, c1;
  int in_h, in_w, out_h, out_w, stride, upsample;
  int pad_lo;       // zero rows / columns in front of the image: 1 (symmetric pad 1) or 0 (VAE-encoder downsample: bottom / right only)
  const void* w;
  long M, N, K;
  const void* bias;
  const void* bvec;
  long ldbvec;
  long rows_per_batch;
  const void* res;
  long ldres;
  int act;
  int geglu;
  float out_scale;
  void* out;
  long ldc;
  long n_split;
  void* out_t;
  long ldt;
  float* ws;
  int full_tiles;   // tiles [0, full_tiles) are computed whole; each later tile is cut into tail_s K-ranges
  int tail_s;
  int tile_bm, tile_bn;
  int kt_per_split;
  int tiles_n;
  long a_rpb, a_bs;
  long lda, ldw;    // gemm_glds_kernel, plain single-source GEMM: row pitches of A / W in elements (= K unless the caller padded them)
  int epi_lds;      // operands / strides allow the LDS-transposed, 16-byte-coalesced epilogue
  const void* a_coef;   // conv slab kernel: GroupNorm coefficients [batch][2][c0 + c1] fp32 (a, d): A' = act(A * a + d) while staging, or NULL
  int a_silu;           // ... with SiLU
  const float* ln_u;    // LayerNorm folded into the GEMM (gemm_glds_kernel<..., LN = true>): W is pre-multiplied by gamma, the kernel takes the
  const float* ln_v;    // row statistics from its own A tiles and the epilogue forms rstd * (acc - mean * u[n]) + v[n]; u, v fp32 [N]
  float ln_eps;
  const float* ln_rows; // LN = 2: precomputed (rstd, -rstd * mean) per row, fp32 [M][2] (tg_layernorm_stats)
  int flags;        // dev experiments (env TG_GEMM_FLAGS): bit 0 = stagger the two co-resident blocks of a CU (low 8 bits = mode,
                    // bits 8.. = delay in ~1 us units), bit 1 = s_setprio(1) around the MFMA chain
  // slab conv PATCH tiles (tg_conv_slab.hip): a 128-pixel tile = patch_np patches of (128 / patch_np >> patch_pwl) rows x (1 << patch_pwl)
  // columns of an in_h x in_w image, patches numbered image-major / patch-row / patch-column.  patch_pwl = 0: tile rows are contiguous tokens.
  int patch_pwl, patch_np;
  // cross-attention epilogue of the LayerNorm-folded to_q projection (gemm_glds_kernel<..., XA>, tg_xattn_epi.h, tg_xq_attn)
  const void* xa_kv;          // [batch][N / 160 tiles][pieces of 1 KiB]: K / V^T MFMA fragments of the tile's heads (tg_xq_kv_pack)
  const float* xa_ip_scale;   // device scalar (IPAttnProcessor.scale) or NULL (1.0)
  int xa_rows_per_batch, xa_tiles_n, xa_L, xa_T;
  int slab_order;   // slab conv work order: 0 tile-major, 1 (column tile, split)-major / row-tile-minor (weight-heavy layers)
  float* gn_part;   // two-wave slab conv, unsplit: GroupNorm partial sums of the output, fp32 [batch][hw / 64][N / gn_cpg][2] (tg_gemm_desc.out_gn_partials), or NULL
  int gn_cpg;       // ... channels per group (divides 80)
};

// token (row of the token-major tensor) of local row `lr` (0..127) of patch tile `tile_m`; see GemmParams::patch_pwl
__device__ __forceinline__ long patch_token(const GemmParams& p, int tile_m, int lr) {
  const int pwl = p.patch_pwl, np = p.patch_np, pp = 128 / np;
  const int k = lr / pp, q = lr - k * pp;
  const int th = pp >> pwl, tpr = p.in_w >> pwl, tpi = (p.in_h / th) * tpr;
  const int g = tile_m * np + k;
  const int img = g / tpi, rem = g - img * tpi;
  const int y0 = (rem / tpr) * th, x0 = (rem - (rem / tpr) * tpr) << pwl;
  return ((long)img * p.in_h + y0 + (q >> pwl)) * p.in_w + x0 + (q & ((1 << pwl) - 1));
}


template <typename T>
__device__ __forceinline__ void epilogue_store4(const GemmParams& p, long m, long n4, float v0, float v1, float v2, float v3) {
  if (m >= p.M || n4 >= p.N) return;
  typedef typename Vec<T>::v4 V4;
  float v[4] = {v0, v1, v2, v3};
  long b = 0;
  if (p.bvec || (p.n_split > 0)) b = m / p.rows_per_batch;
  if (p.bias) {
    V4 t = *reinterpret_cast<const V4*>(reinterpret_cast<const T*>(p.bias) + n4);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += to_f32<T>(t[j]);
  }
  if (p.bvec) {
    V4 t = *reinterpret_cast<const V4*>(reinterpret_cast<const T*>(p.bvec) + b * p.ldbvec + n4);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += to_f32<T>(t[j]);
  }
  if (p.res) {
    V4 t = *reinterpret_cast<const V4*>(reinterpret_cast<const T*>(p.res) + m * p.ldres + n4);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += to_f32<T>(t[j]);
  }
  if (p.act != TG_ACT_NONE) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], p.act);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] *= p.out_scale;
  if (p.n_split > 0 && n4 >= p.n_split) {
    T* o = reinterpret_cast<T*>(p.out_t);
    long tok = m - b * p.rows_per_batch;
    long nt = p.N - p.n_split;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[(b * nt + (n4 + j - p.n_split)) * p.ldt + tok] = from_f32<T>(v[j]);
  } else {
    V4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = from_f32<T>(v[j]);
    *reinterpret_cast<V4*>(reinterpret_cast<T*>(p.out) + m * p.ldc + n4) = o;
  }
}

// EXPERIMENT (TG_GEMM_FLAGS bit 0): the two blocks that share a CU start together and have identical work, so they sit in
// their K loops together (each with half the matrix pipe) and in their epilogues together (matrix pipe idle).  Delay ONE of
// the two first-round blocks of every CU by about half a tile period so that one block's epilogue overlaps the other's K
// loop; later rounds inherit the offset (a finished block is replaced at once).  CU identity from the hardware id registers,
// arrival parity from a never-reset global counter (consecutive arrivals on one CU differ in parity).  Speed only.
__device__ unsigned int tg_cu_arrivals[2048];
__device__ __forceinline__ void stagger_first_round(int flags, char* smem_base) {
  if (!(flags & 1) || gridDim.x < 512 || blockIdx.x >= 512) return;
  int* dec = reinterpret_cast<int*>(smem_base);
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
    const unsigned key = ((xcc & 7u) << 8) | ((hw >> 8) & 0xffu);
    *dec = (int)(atomicAdd(&tg_cu_arrivals[key], 1u) & 1u);
  }
  __syncthreads();
  const int late = *dec;
  __syncthreads();
  if (late) {
    const int units = (flags >> 8) & 0xff;
    for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(32);         // ~2048 cycles ~ 1 us each
  }
}

// XCD-aware tile order (speed only, never correctness): the dispatcher places block b on XCD b % 8, each XCD has a
// private 4 MiB L2.  With the natural order the tiles that share an activation row-slab (same tile_m, different
// tile_n) land on different XCDs and every L2 fetches the slab again from the fabric (measured: 44 % L2 miss rate,
// ~3.9 TB/s of miss traffic on the 64x64 320->320 conv).  Remap so that XCD x owns a CONTIGUOUS chunk of logical
// tiles (bijective for any grid size): neighbours in (tile_m, tile_n) order run on the same L2 at the same time.
__device__ __forceinline__ int xcd_chunked_block_id(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7;
  const int xcd = bid & 7;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + (bid >> 3);
}

// Whole-wave-tile epilogue.  All bias / per-batch vector / residual loads of one 32-token row block are issued
// back to back BEFORE any of them is consumed (one latency exposure per row block instead of one per 4 outputs),
// then activation / scale / 8-byte stores.  lane&31 = token row, 4 consecutive registers = 4 consecutive channels.
// EPI: 0 = linear epilogue only (bias / per-batch vector / residual / scale: every conv and most projections),
// 1 = generic (activations, GEGLU), 2 = GEGLU only.  The activation code (erf polynomials, exp) is ~80 % of the
// kernel's instructions; leaving it out of the kernels that never run it is worth ~6 % at K = 320 (code size).
// J0 / JN: the window [J0, J0 + JN) of the wave tile's TN 32-column tiles this call handles (big wave tiles run the
// epilogue in column chunks so that the LDS bounce stays small); n_base is the column of tile 0, lane offset included.
// LN: LayerNorm-folded projection (gemm_glds_kernel<..., LN>): the kernel has already turned every accumulator into
// rstd * (acc - mean * u[n]); the epilogue adds the fp32 vector v[n] = sum_k beta[k] W[n, k] + bias[n] where the bias would go.
// Rows of a wave tile -> tokens.  Default: row q (0 .. 32 TM - 1) of the wave tile is token m_wave + mstride * (q / 32) + q % 32 (mstride = 32:
// contiguous rows; the image width for the slab conv's 32-pixel patch rows).  PR = true (template flag of the epilogue functions; slab conv
// patch rows of 1 << pwl <= 16 pixels only, so that every other kernel keeps its address arithmetic): m_wave + (q >> pwl) * in_w + (q & ((1 << pwl) - 1)).
__device__ __forceinline__ long patch_row_token(const GemmParams& p, long m_wave, int q, int pwl) {
  return m_wave + (long)(q >> pwl) * p.in_w + (q & ((1 << pwl) - 1));
}

template <typename T, int TM, int TN, int EPI, int J0 = 0, int JN = TN, bool LN = false, bool PR = false>
__device__ __forceinline__ void epilogue_tile(const GemmParams& p, f32x16 (&acc)[TM][TN], long m_base, long n_base, int part,
                                              long pm0, long pn0, long mstride = 32, int pwl = 0) {
  typedef typename Vec<T>::v4 V4;
  // m_base = m_wave + (lane & 31)
  auto row_m = [&](int i) -> long {
    if constexpr (PR) { const int l31e = (int)threadIdx.x & 31; return patch_row_token(p, m_base - l31e, 32 * i + l31e, pwl); }
    else return m_base + mstride * i;
  };
  if constexpr (LN) {
    // (the caller already formed rstd * (acc - mean * u) in place) + v[n]; the direct path is the cold one: V^T columns
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = J0; j < J0 + JN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const long n4 = n_base + 32 * j + 8 * g;
          f32x4 v4 = {0.f, 0.f, 0.f, 0.f};
          if (n4 < p.N) v4 = *reinterpret_cast<const f32x4*>(p.ln_v + n4);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] += v4[e];
        }
  }
  if (part >= 0) {
    // K-split tail tile: fp32 partial in tile-local layout ws[part][tile_bm][tile_bn]; the reduce kernel sums the
    // tail_s partials of the tile in a fixed order and applies the epilogue
    float* wsp = p.ws + (long)part * p.tile_bm * p.tile_bn;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const long lr = m_base + 32 * i - pm0;
#pragma unroll
      for (int j = J0; j < J0 + JN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const long lc = n_base + 32 * j + 8 * g - pn0;
          f32x4 o = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          *reinterpret_cast<f32x4*>(wsp + lr * p.tile_bn + lc) = o;
        }
    }
    return;
  }
  if constexpr (JN % 2 == 0 && J0 % 2 == 0 && EPI != 0) {
    if (EPI == 2 || p.geglu) {
      // fused GEGLU (models/attention.py:337-338): W rows are packed [a(32) ; gate(32)] per 64-column group, so this
      // wave holds a[c] in tile 2q and gate[c] in tile 2q + 1 for the same 32 channels, in the same lane/register:
      // out[m, c] = (a + bias_a) * gelu(gate + bias_g), written to a [M, N/2] tensor — the [M, N] pre-activation
      // never exists in HBM.
      const T* biasp = reinterpret_cast<const T*>(p.bias);
      const long hi4 = n_base & 31;                 // 4 * (lane >> 5)
#pragma unroll
      for (int jq = J0; jq < J0 + JN; jq += 2) {
        const long nA = n_base - hi4 + 32 * jq;     // first packed column of the a-block (multiple of 64)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const long m = row_m(i);
          if (m >= p.M) continue;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const long na = nA + hi4 + 8 * g;         // packed column of a; gate sits 32 further
            if (na + 32 >= p.N) continue;
            V4 ba, bg;
#pragma unroll
            for (int e = 0; e < 4; ++e) { ba[e] = from_f32<T>(0.f); bg[e] = from_f32<T>(0.f); }
            if (biasp != nullptr) {
              ba = *reinterpret_cast<const V4*>(biasp + na);
              bg = *reinterpret_cast<const V4*>(biasp + na + 32);
            }
            V4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float a = acc[i][jq][4 * g + e] + to_f32<T>(ba[e]);
              const float gt = acc[i][jq + 1][4 * g + e] + to_f32<T>(bg[e]);
              o[e] = from_f32<T>(a * gelu_erf_f(gt) * p.out_scale);
            }
            *reinterpret_cast<V4*>(reinterpret_cast<T*>(p.out) + m * p.ldc + (nA >> 1) + hi4 + 8 * g) = o;
          }
        }
      }
      return;
    }
  }
  const T* biasp = reinterpret_cast<const T*>(p.bias);
  const T* bvecp = reinterpret_cast<const T*>(p.bvec);
  const T* resp = reinterpret_cast<const T*>(p.res);
  V4 zero4;
#pragma unroll
  for (int e = 0; e < 4; ++e) zero4[e] = from_f32<T>(0.f);
  V4 bias4[TN][4];
#pragma unroll
  for (int j = J0; j < J0 + JN; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const long n4 = n_base + 32 * j + 8 * g;
      bias4[j][g] = (biasp != nullptr && n4 < p.N) ? *reinterpret_cast<const V4*>(biasp + n4) : zero4;
    }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long m = row_m(i);
    const bool m_ok = m < p.M;
    long b = 0;
    if (bvecp != nullptr || p.n_split > 0) b = (m_ok ? m : 0) / p.rows_per_batch;
    V4 add4[TN][4], res4[TN][4];
#pragma unroll
    for (int j = J0; j < J0 + JN; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const long n4 = n_base + 32 * j + 8 * g;
        const bool ok = m_ok && n4 < p.N;
        add4[j][g] = (bvecp != nullptr && ok) ? *reinterpret_cast<const V4*>(bvecp + b * p.ldbvec + n4) : zero4;
        res4[j][g] = (resp != nullptr && ok) ? *reinterpret_cast<const V4*>(resp + m * p.ldres + n4) : zero4;
      }
#pragma unroll
    for (int j = J0; j < J0 + JN; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const long n4 = n_base + 32 * j + 8 * g;
        if (!(m_ok && n4 < p.N)) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] = acc[i][j][4 * g + e] + to_f32<T>(bias4[j][g][e]) + to_f32<T>(add4[j][g][e]) + to_f32<T>(res4[j][g][e]);
        if constexpr (EPI == 1) {
          if (p.act != TG_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= p.out_scale;
        if (p.n_split > 0 && n4 >= p.n_split) {
          T* o = reinterpret_cast<T*>(p.out_t);
          const long tok = m - b * p.rows_per_batch;
          const long nt = p.N - p.n_split;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[(b * nt + (n4 + e - p.n_split)) * p.ldt + tok] = from_f32<T>(v[e]);
        } else {
          V4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(v[e]);
          *reinterpret_cast<V4*>(reinterpret_cast<T*>(p.out) + m * p.ldc + n4) = o;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------------------
// LDS-transposed epilogue.  In the MFMA accumulator layout a lane owns ONE token row and 4 consecutive channels per
// register quad, so the direct epilogue above issues 8-byte stores to 32 different 128-byte lines per instruction
// (and the same pattern for residual loads): 16 K line accesses per CU per tile round, ~8.7 us of fixed cost per
// 128x128 tile (K sweep in profiles/r1_gemm_findings.md) — more than the K loop itself when K <= 640.  Here each wave
// bounces its fp32 accumulators through a private LDS scratch (the operand stages are dead after the K loop's last
// barrier) and comes back with lane = (row, 8-channel piece): bias / per-batch vector / residual are 16-byte loads,
// the store is 16 bytes per lane, 8 full 128-byte lines per wave-instruction.  Arithmetic and its order are
// unchanged (fp32: acc + bias + bvec + res, activation, scale, one rounding), so results are bit-identical to the
// direct epilogue.  Columns that go to the transposed output (out_t, lane = token is already coalesced there) and
// split-K partials keep the direct path.
// Row loop of the LDS-transposed epilogue (see epilogue_tile_lds): straight-line per 32-row half — all residual /
// per-batch-vector loads, then the 8 LDS writes, then ALL LDS reads of the half, then the arithmetic and the stores
// (only the store is predicated on the row bound).  fp32: ((acc + bias) + bvec) + res, activation, * scale, one rounding;
// x + 0 and x * 1 are exact, so this rounds the same value as the direct epilogue.
template <typename T, int TM, int TN, int EPI, bool HAS_ADD, bool HAS_RES, int J0 = 0, int JN = TN, bool LN = false, bool PR = false>
__device__ __forceinline__ void epilogue_rows_lds(const GemmParams& p, f32x16 (&acc)[TM][TN], long m_wave, long n_wave, int lane,
                                                  float* scr, long mstride = 32, int pwl = 0) {
  typedef typename Vec<T>::v8 V8;
  constexpr int W = JN * 32, RS = W + 4, P = W / 8, RPP = 64 / P, NPASS = 32 / RPP;
  const int l31 = lane & 31, hi = lane >> 5;
  const int c = lane % P, r0 = lane / P;
  const long n = n_wave + J0 * 32 + c * 8;
  const bool n_ok = n < p.N;
  T* outp = reinterpret_cast<T*>(p.out);
  const T* biasp = reinterpret_cast<const T*>(p.bias);
  const T* bvecp = reinterpret_cast<const T*>(p.bvec);
  const T* resp = reinterpret_cast<const T*>(p.res);
  const long nc = n_ok ? n : 0;                      // clamped column: loads stay in bounds, the store is predicated
  float bias_f[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias_f[e] = 0.f;
  if (biasp != nullptr) {
    const V8 b8 = *reinterpret_cast<const V8*>(biasp + nc);
#pragma unroll
    for (int e = 0; e < 8; ++e) bias_f[e] = to_f32<T>(b8[e]);
  }
  if constexpr (LN) {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(p.ln_v + nc), v1 = *reinterpret_cast<const f32x4*>(p.ln_v + nc + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { bias_f[e] += v0[e]; bias_f[4 + e] += v1[e]; }
  }
  const float scale = p.out_scale;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long m_first = m_wave + mstride * i + r0;      // mstride: token distance between the wave tile's 32-row blocks (32 = contiguous rows;
    // the image width for the slab conv's 32-pixel patch rows); PR: token of this lane's row in pass `it` from the patch-row mapping
    auto mrow = [&](int it) -> long {
      if constexpr (PR) return patch_row_token(p, m_wave, 32 * i + it * RPP + r0, pwl);
      else return m_first + it * RPP;
    };
    V8 add8[NPASS], res8[NPASS];
    if constexpr (HAS_ADD) {
#pragma unroll
      for (int it = 0; it < NPASS; ++it) {
        long m = mrow(it);
        if (m >= p.M) m = p.M - 1;
        add8[it] = *reinterpret_cast<const V8*>(bvecp + (m / p.rows_per_batch) * p.ldbvec + nc);
      }
    }
    if constexpr (HAS_RES) {
#pragma unroll
      for (int it = 0; it < NPASS; ++it) {
        long m = mrow(it);
        if (m >= p.M) m = p.M - 1;
        res8[it] = *reinterpret_cast<const V8*>(resp + m * p.ldres + nc);
      }
    }
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 o = {acc[i][J0 + j][4 * g], acc[i][J0 + j][4 * g + 1], acc[i][J0 + j][4 * g + 2], acc[i][J0 + j][4 * g + 3]};
        *reinterpret_cast<f32x4*>(scr + l31 * RS + 32 * j + 8 * g + 4 * hi) = o;
      }
    __builtin_amdgcn_wave_barrier();
    f32x4 lo[NPASS], hi4[NPASS];
#pragma unroll
    for (int it = 0; it < NPASS; ++it) {
      lo[it] = *reinterpret_cast<const f32x4*>(scr + (it * RPP + r0) * RS + c * 8);
      hi4[it] = *reinterpret_cast<const f32x4*>(scr + (it * RPP + r0) * RS + c * 8 + 4);
    }
    __builtin_amdgcn_wave_barrier();
    T* op = outp + m_first * p.ldc + n;
#pragma unroll
    for (int it = 0; it < NPASS; ++it) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = lo[it][e] + bias_f[e]; v[4 + e] = hi4[it][e] + bias_f[4 + e]; }
      if constexpr (HAS_ADD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += to_f32<T>(add8[it][e]);
      }
      if constexpr (HAS_RES) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += to_f32<T>(res8[it][e]);
      }
      if constexpr (EPI == 1) {
        if (p.act != TG_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], p.act);
        }
      }
      V8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(v[e] * scale);
      if constexpr (PR) {
        if (const long m = mrow(it); m < p.M && n_ok) *reinterpret_cast<V8*>(outp + m * p.ldc + n) = o;
      } else {
        if (m_first + it * RPP < p.M && n_ok) *reinterpret_cast<V8*>(op + (long)it * RPP * p.ldc) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Register-only epilogue with 16-byte rows ("reg16").  The LDS bounce above costs two wave barriers and an LDS round trip per
// 32-row block and serialises the blocks of a wave tile on one scratch buffer; in the MFMA accumulator layout the two halves
// of a wave already hold ADJACENT channel quads of the SAME token row (lane l: channels 8g + 0..3, lane l + 32: 8g + 4..7), so
// one v_permlane32_swap per packed dword between the quads g and g + 1 turns them into 8 contiguous channels per lane
// (guide T21): 16-byte stores, 32 contiguous bytes per token row and instruction, no LDS, no barrier, and every 32 x 32 tile
// is independent of the others (the compiler pipelines residual loads, arithmetic and stores across tiles).  The residual
// is loaded in the SAME 16-byte layout and brought into accumulator layout by the same swap (it is an involution), so the
// fp32 arithmetic — ((acc + bias) + res), activation, * scale, ONE rounding — and therefore every output bit equals the LDS
// and the direct epilogue's.  Per-batch vectors (time-embedding adds of the convs) keep the LDS path.
__device__ __forceinline__ void lane_half_swap(unsigned& a, unsigned& b) {
  // a: lanes 32-63 <-> b: lanes 0-31 (the other two halves stay); s_nop covers the VALU-write -> permlane-read hazard
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

template <typename T>
__device__ __forceinline__ void unpack4(unsigned d0, unsigned d1, float (&f)[4]) {
  typedef typename Vec<T>::v4 V4;
  const u32x2 d = {d0, d1};
  const V4 v = __builtin_bit_cast(V4, d);
#pragma unroll
  for (int e = 0; e < 4; ++e) f[e] = to_f32<T>(v[e]);
}
template <typename T>
__device__ __forceinline__ u32x2 pack4(const float (&f)[4]) {
  typedef typename Vec<T>::v4 V4;
  V4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = from_f32<T>(f[e]);
  return __builtin_bit_cast(u32x2, v);
}

// one wave tile; requires p.epi_lds (16-byte alignment of out / bias / res and their pitches, N % 8 == 0) and p.bvec == NULL
template <typename T, int TM, int TN, int EPI>
__device__ __forceinline__ void epilogue_tile_reg16(const GemmParams& p, f32x16 (&acc)[TM][TN], long m_wave, long n_wave, int lane,
                                                    long pm0, long pn0) {
  typedef typename Vec<T>::v4 V4;
  const int l31 = lane & 31, hi = lane >> 5;
  T* outp = reinterpret_cast<T*>(p.out);
  const T* biasp = reinterpret_cast<const T*>(p.bias);
  const T* resp = reinterpret_cast<const T*>(p.res);
  const float scale = p.out_scale;
  if constexpr (TN % 2 == 0 && EPI != 0) {
    if (EPI == 2 || p.geglu) {
#pragma unroll
      for (int jq = 0; jq < TN; jq += 2) {
        const long nq = n_wave + 32 * jq;                 // first packed column of this (a, gate) pair; output column nq / 2
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          float ba[2][4], bg[2][4];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const long na = nq + 16 * gp + 8 * q + 4 * hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) { ba[q][e] = 0.f; bg[q][e] = 0.f; }
            if (biasp != nullptr && na + 32 < p.N) {
              const V4 a4 = *reinterpret_cast<const V4*>(biasp + na), g4 = *reinterpret_cast<const V4*>(biasp + na + 32);
#pragma unroll
              for (int e = 0; e < 4; ++e) { ba[q][e] = to_f32<T>(a4[e]); bg[q][e] = to_f32<T>(g4[e]); }
            }
          }
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const long m = m_wave + 32 * i + l31;
            float v[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float a = acc[i][jq][8 * gp + 4 * q + e] + ba[q][e];
                const float gt = acc[i][jq + 1][8 * gp + 4 * q + e] + bg[q][e];
                v[q][e] = a * gelu_erf_f(gt) * scale;
              }
            const u32x2 p0 = pack4<T>(v[0]), p1 = pack4<T>(v[1]);
            unsigned a0 = p0[0], a1 = p0[1], b0 = p1[0], b1 = p1[1];
            lane_half_swap(a0, b0);
            lane_half_swap(a1, b1);
            const long nf = (nq >> 1) + 16 * gp + 8 * hi;
            const u32x4 o = {a0, a1, b0, b1};
            if (m < p.M && nq + 16 * gp + 8 * hi + 32 < p.N) *reinterpret_cast<u32x4*>(outp + m * p.ldc + nf) = o;
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const long nt = n_wave + 32 * j;
    if (p.n_split > 0 && nt >= p.n_split) {
      // columns of the transposed output (V^T): lane = token is already the coalesced layout there
      epilogue_tile<T, TM, TN, EPI>(p, acc, m_wave + l31, n_wave + 4 * hi, -1, pm0, pn0);       // (window version below)
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const long nt = n_wave + 32 * j;
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      const long nf = nt + 16 * gp + 8 * hi;              // this lane's 8 output channels after the swap
      const bool n_ok = nf < p.N;
      float bq[2][4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const long na = nt + 16 * gp + 8 * q + 4 * hi;    // this lane's accumulator channels of quad 2 gp + q
#pragma unroll
        for (int e = 0; e < 4; ++e) bq[q][e] = 0.f;
        if (biasp != nullptr && na < p.N) {
          const V4 b4 = *reinterpret_cast<const V4*>(biasp + na);
#pragma unroll
          for (int e = 0; e < 4; ++e) bq[q][e] = to_f32<T>(b4[e]);
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const long m = m_wave + 32 * i + l31;
        const bool ok = m < p.M && n_ok;
        float r[2][4];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) r[q][e] = 0.f;
        if (resp != nullptr) {
          u32x4 r4 = {0u, 0u, 0u, 0u};
          if (ok) r4 = *reinterpret_cast<const u32x4*>(resp + m * p.ldres + nf);
          unsigned a0 = r4[0], a1 = r4[1], b0 = r4[2], b1 = r4[3];
          lane_half_swap(a0, b0);
          lane_half_swap(a1, b1);
          unpack4<T>(a0, a1, r[0]);
          unpack4<T>(b0, b1, r[1]);
        }
        float v[2][4];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = (acc[i][j][8 * gp + 4 * q + e] + bq[q][e]) + r[q][e];
            if constexpr (EPI == 1) {
              if (p.act != TG_ACT_NONE) x = apply_act(x, p.act);
            }
            v[q][e] = x * scale;
          }
        const u32x2 p0 = pack4<T>(v[0]), p1 = pack4<T>(v[1]);
        unsigned a0 = p0[0], a1 = p0[1], b0 = p1[0], b1 = p1[1];
        lane_half_swap(a0, b0);
        lane_half_swap(a1, b1);
        const u32x4 o = {a0, a1, b0, b1};
        if (ok) *reinterpret_cast<u32x4*>(outp + m * p.ldc + nf) = o;
      }
    }
  }
}

// one column chunk [J0, J0 + JN) of the wave tile through the LDS bounce (or the direct path where the bounce does not apply)
template <typename T, int TM, int TN, int EPI, int J0, int JN, bool LN = false, bool PR = false>
__device__ __forceinline__ void epilogue_chunk_lds(const GemmParams& p, f32x16 (&acc)[TM][TN], long m_wave, long n_wave, int lane,
                                                   float* scr, int part, long pm0, long pn0, long mstride = 32, int pwl = 0) {
  typedef typename Vec<T>::v4 V4;
  typedef typename Vec<T>::v8 V8;
  const int l31 = lane & 31, hi = lane >> 5;
  if (part >= 0 || !p.epi_lds || (p.n_split > 0 && n_wave + J0 * 32 >= p.n_split)) {
    epilogue_tile<T, TM, TN, EPI, J0, JN, LN, PR>(p, acc, m_wave + l31, n_wave + 4 * hi, part, pm0, pn0, mstride, pwl);
    return;
  }
  T* outp = reinterpret_cast<T*>(p.out);
  const T* biasp = reinterpret_cast<const T*>(p.bias);
  if constexpr (JN % 2 == 0 && J0 % 2 == 0 && EPI != 0) {
    if (EPI == 2 || p.geglu) {
      constexpr int RS = 36;
      const int c = lane & 3, r0 = lane >> 2;       // 4 pieces x 16 rows per pass over the [32][32] result
      const float scale = p.out_scale;
#pragma unroll
      for (int jq = J0; jq < J0 + JN; jq += 2) {
        const long nq = n_wave + 32 * jq;           // first packed column of this (a, gate) pair
        // the 2 x 4 bias quads of this lane's channels do not depend on the row block: loaded once (fp32), straight-line body
        float baf[4][4], bgf[4][4];
        bool gok[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const long na = nq + 4 * hi + 8 * g;
          gok[g] = na + 32 < p.N;
          V4 ba, bg;
#pragma unroll
          for (int e = 0; e < 4; ++e) { ba[e] = from_f32<T>(0.f); bg[e] = from_f32<T>(0.f); }
          if (biasp != nullptr && gok[g]) {
            ba = *reinterpret_cast<const V4*>(biasp + na);
            bg = *reinterpret_cast<const V4*>(biasp + na + 32);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) { baf[g][e] = to_f32<T>(ba[e]); bgf[g][e] = to_f32<T>(bg[e]); }
        }
        if constexpr (LN) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const long na = gok[g] ? nq + 4 * hi + 8 * g : 0;
            const f32x4 va = *reinterpret_cast<const f32x4*>(p.ln_v + na), vg = *reinterpret_cast<const f32x4*>(p.ln_v + na + 32);
#pragma unroll
            for (int e = 0; e < 4; ++e) { baf[g][e] += va[e]; bgf[g][e] += vg[e]; }
          }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float a = acc[i][jq][4 * g + e] + baf[g][e];
              const float gt = acc[i][jq + 1][4 * g + e] + bgf[g][e];
              o[e] = gok[g] ? a * gelu_erf_f(gt) * scale : 0.f;
            }
            *reinterpret_cast<f32x4*>(scr + l31 * RS + 8 * g + 4 * hi) = o;
          }
          __builtin_amdgcn_wave_barrier();
          f32x4 lo[2], hi4[2];
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            lo[it] = *reinterpret_cast<const f32x4*>(scr + (it * 16 + r0) * RS + c * 8);
            hi4[it] = *reinterpret_cast<const f32x4*>(scr + (it * 16 + r0) * RS + c * 8 + 4);
          }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const long m = m_wave + mstride * i + it * 16 + r0;
            V8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = from_f32<T>(lo[it][e]); o[4 + e] = from_f32<T>(hi4[it][e]); }
            if (m < p.M && nq + c * 8 + 32 < p.N) *reinterpret_cast<V8*>(outp + m * p.ldc + (nq >> 1) + c * 8) = o;
          }
        }
      }
      return;
    }
  }
  const T* bvecp = reinterpret_cast<const T*>(p.bvec);
  const T* resp = reinterpret_cast<const T*>(p.res);
  // one straight-line instance of the row loop per (per-batch vector?, residual?) combination: with the wave-uniform
  // branches inside the loop every pass was its own basic block and the compiler exposed one LDS / load latency per
  // pass (in-kernel s_memtime: ~7200 cycles per 128x128 tile against ~1800 per K-tile)
  if constexpr (LN) {
    // the LayerNorm-fused projections carry no per-batch vector and no residual (QKV, to_q, FF1)
    epilogue_rows_lds<T, TM, TN, EPI, false, false, J0, JN, true>(p, acc, m_wave, n_wave, lane, scr, mstride);
  } else if (bvecp != nullptr) {
    if (resp != nullptr) epilogue_rows_lds<T, TM, TN, EPI, true, true, J0, JN, false, PR>(p, acc, m_wave, n_wave, lane, scr, mstride, pwl);
    else epilogue_rows_lds<T, TM, TN, EPI, true, false, J0, JN, false, PR>(p, acc, m_wave, n_wave, lane, scr, mstride, pwl);
  } else {
    if (resp != nullptr) epilogue_rows_lds<T, TM, TN, EPI, false, true, J0, JN, false, PR>(p, acc, m_wave, n_wave, lane, scr, mstride, pwl);
    else epilogue_rows_lds<T, TM, TN, EPI, false, false, J0, JN, false, PR>(p, acc, m_wave, n_wave, lane, scr, mstride, pwl);
  }
}

// Whole wave tile.  Up to two 32-column tiles go through the bounce in one piece (the 128x128 / 64x64 kernels: unchanged);
// wider wave tiles (the big-tile kernels: 5 or 4 tiles) run in 64-column chunks (+ one 32-column rest), so the per-wave
// scratch stays 32 x 68 floats and every store instruction still covers whole 128-byte rows.
template <typename T, int TM, int TN, int EPI, bool LN = false, bool PR = false>
__device__ __forceinline__ void epilogue_tile_lds(const GemmParams& p, f32x16 (&acc)[TM][TN], long m_wave, long n_wave, int lane,
                                                  float* scr, int part, long pm0, long pn0, long mstride = 32, int pwl = 0) {
  if constexpr (TN <= 2) {
    epilogue_chunk_lds<T, TM, TN, EPI, 0, TN, LN, PR>(p, acc, m_wave, n_wave, lane, scr, part, pm0, pn0, mstride, pwl);
  } else {
    // (LN: the 160-column wave tiles of the LayerNorm-folded projections, round 5; the fold's fp32 vector v rides where the bias does, per chunk)
    epilogue_chunk_lds<T, TM, TN, EPI, 0, 2, LN, PR>(p, acc, m_wave, n_wave, lane, scr, part, pm0, pn0, mstride, pwl);
    if constexpr (TN >= 4) epilogue_chunk_lds<T, TM, TN, EPI, 2, 2, LN, PR>(p, acc, m_wave, n_wave, lane, scr, part, pm0, pn0, mstride, pwl);
    if constexpr (TN == 5) epilogue_chunk_lds<T, TM, TN, EPI, 4, 1, LN, PR>(p, acc, m_wave, n_wave, lane, scr, part, pm0, pn0, mstride, pwl);
    if constexpr (TN == 3) epilogue_chunk_lds<T, TM, TN, EPI, 2, 1, LN, PR>(p, acc, m_wave, n_wave, lane, scr, part, pm0, pn0, mstride, pwl);
    static_assert(TN <= 5, "wave tiles wider than 160 columns are not instantiated");
  }
}

// out-of-range rows / conv padding of the LDS-DMA gathers read from this page of zeros
__device__ __attribute__((aligned(256))) unsigned char tg_zero_page[256];

}  // namespace

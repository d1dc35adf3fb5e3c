// Cross-attention as the EPILOGUE of the LayerNorm-folded to_q projection (round 5; gemm_glds_kernel<..., XA = head dim>, launched by tg_xq_attn).
//
// Inner levels of the UNet (32 x 32: C = 640 = 8 heads x 80; 16 x 16: C = 1280 = 8 heads x 160), BasicTransformerBlock's second sub-block
// (models/attention.py:206-224; processors ip_adapter/attention_processor.py:282-393 AttnProcessor, :396-553 IPAttnProcessor):
//     q = norm2(x) Wq^T ;  O = softmax(s q Kt^T) Vt + w softmax(s q Kip^T) Vip        (77 text keys, T <= 16 image keys, two independent softmaxes)
// was three launches: LayerNorm-folded to_q GEMM (writes q), the flash attention kernel (reads q, 81 keys: 28 / 23 us for 7 us of memory time), to_out.
// Here the 128 x 160 tile of the to_q GEMM (four waves of 32 tokens x 160 channels = two heads of 80 or one head of 160) keeps q in its accumulators:
//   * the W rows are read with bits 2 / 3 of the MFMA row swapped, so 8 consecutive accumulator registers are 8 consecutive channels: the rounded q is
//     the B operand of S^T = K q^T as it stands (10 fragments of 16 channels);
//   * K rows are packed with the same swap on the KEY index, so the probabilities' accumulator registers are, 8 by 8, the B operand of O^T += V^T P^T;
//   * K / V^T of the tile's heads come as pre-packed 1-KiB MFMA fragments (tg_xq_kv_pack, once per conditioning) through LDS (the operand stages are dead);
//   * softmax: lane = query, the scores of a query sit in two lanes (hi halves): one cross-half exchange per max / sum; exp2 with the scale folded into
//     Wq (the host packs s log2(e) Wq); probabilities are normalised (and the image segment weighted by the device-resident IP scale) BEFORE they are
//     rounded for the PV product — what the reference's half-precision softmax output does;
//   * O^T accumulates in the standard accumulator layout (V^T rows unswapped), two heads of 80 into one 160-channel tile (the shared 32-row block takes
//     zero rows from the other head), and leaves through the GEMM's LDS-transposed epilogue: q and the attention launch never exist.
#pragma once
#include "tg_gemm_common.h"

namespace {

__device__ __forceinline__ int xa_swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

// pieces (1 KiB fragments) of one head's K / V^T set: K: 4 key blocks (3 text, 1 image) x D / 16 k-steps; V^T: DBH row blocks x 7 key k-steps (6 text, 1 image)
template <int D> struct XaGeom {
  static constexpr int KS = D / 16;                 // q / K k-steps per head
  static constexpr int DBH = D == 80 ? 3 : 5;       // 32-row blocks of O^T a head touches
  static constexpr int HPT = 160 / D;               // heads per 160-column tile
  static constexpr int NPH = 4 * KS + 7 * DBH;      // pieces per head: 41 / 75
  static constexpr int NPT = HPT * NPH;             // pieces per tile: 82 / 75
};

__device__ __forceinline__ float xa_other_half(float v) {
  // value of the same query in the other lane half (lane ^ 32)
  return __shfl_xor(v, 32, 64);
}

template <typename T, int D>
__device__ __forceinline__ void xattn_epilogue(const GemmParams& p, f32x16 (&acc)[1][5], char* smem, int wave, int lane, long m0, long n0, int tile_n) {
  typedef typename Vec<T>::v8 V8;
  typedef XaGeom<D> G;
  const int l31 = lane & 31, hi = lane >> 5;
  // ---- q = acc + v (the LayerNorm fold's fp32 vector, permuted like the W rows), rounded: B fragments of the 10 k-steps of the tile
  V8 qb[10];
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(p.ln_v + n0 + 32 * j + 16 * (g >> 1) + 8 * hi + 4 * (g & 1));
#pragma unroll
      for (int e = 0; e < 4; ++e) qb[2 * j + (g >> 1)][4 * (g & 1) + e] = from_f32<T>(acc[0][j][4 * g + e] + v4[e]);
    }
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
  const long bi = m0 / p.xa_rows_per_batch;                      // the tile's batch item (uniform: rows_per_batch % 128 == 0)
  const char* blob = reinterpret_cast<const char*>(p.xa_kv) + (bi * p.xa_tiles_n + tile_n) * (long)(G::NPT * 1024);
  const float ipw = p.xa_T > 0 ? (p.xa_ip_scale != nullptr ? *p.xa_ip_scale : 1.0f) : 0.f;
  const int L = p.xa_L, Tn = p.xa_T;
#pragma unroll
  for (int h = 0; h < G::HPT; ++h) {
    __syncthreads();                                            // LDS free: K loop / LayerNorm vectors / the previous head's fragments are done with
    for (int q = wave; q < G::NPH; q += 4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(blob + (long)(h * G::NPH + q) * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(smem + q * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const char* kf = smem + lane * 16;
    // ---- S^T[key block][query] = K q^T over the head's k-steps
    f32x16 s[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks)
        s[kb] = mfma32(*reinterpret_cast<const V8*>(kf + (kb * G::KS + ks) * 1024), qb[h * G::KS + ks], s[kb]);
    }
    // ---- two softmaxes (text keys: blocks 0..2, image keys: block 3).  key of register r of block kb: 32 kb + 16 (g >> 1) + 8 hi + 4 (g & 1) + e
    float mt = -INFINITY, mi = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int g = r >> 2, e = r & 3;
        const int key = (kb < 3 ? 32 * kb : 0) + 16 * (g >> 1) + 8 * hi + 4 * (g & 1) + e;
        const bool ok = kb < 3 ? key < L : key < Tn;
        s[kb][r] = ok ? s[kb][r] : -INFINITY;
        if (kb < 3) mt = fmaxf(mt, s[kb][r]); else mi = fmaxf(mi, s[kb][r]);
      }
    mt = fmaxf(mt, xa_other_half(mt));
    mi = fmaxf(mi, xa_other_half(mi));
    if (Tn <= 0) mi = 0.f;
    float lt = 0.f, li = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pe = __builtin_amdgcn_exp2f(s[kb][r] - (kb < 3 ? mt : mi));      // masked keys: exp2(-inf) = 0
        s[kb][r] = pe;
        if (kb < 3) lt += pe; else li += pe;
      }
    lt += xa_other_half(lt);
    li += xa_other_half(li);
    const float wt = 1.0f / lt, wi = Tn > 0 ? ipw / li : 0.f;
    // ---- O^T += V^T P^T: P fragments = 8 consecutive registers (16 keys per k-step: block kb, half s'), normalised before rounding
    const char* vf = kf + 4 * G::KS * 1024;
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) {
      const int kb = ks < 6 ? ks >> 1 : 3, sp = ks < 6 ? ks & 1 : 0;
      const float wgt = ks < 6 ? wt : wi;
      V8 pb;
#pragma unroll
      for (int e = 0; e < 8; ++e) pb[e] = from_f32<T>(s[kb][8 * sp + e] * wgt);
#pragma unroll
      for (int db = 0; db < G::DBH; ++db) {
        const int gb = D == 80 ? 2 * h + db : db;               // global 32-row block of the tile's O^T
        acc[0][gb] = mfma32(*reinterpret_cast<const V8*>(vf + (db * 7 + ks) * 1024), pb, acc[0][gb]);
      }
    }
  }
  __syncthreads();                                              // every wave is done with the fragments: the epilogue's scratch may overwrite them
}

}  // namespace

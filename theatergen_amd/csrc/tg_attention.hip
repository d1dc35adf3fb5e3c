// Flash-style fused attention on MFMA for gfx950 (include/theatergen_hip.h: tg_attention, tg_attn_probs).
//
//   O = softmax(s Q K0^T) V0  +  w1 * softmax(s Q K1^T) V1        (segment 1 optional, len1 <= 64)
//
// No [B*h, N, Lk] probability tensor ever reaches HBM (the reference materialises it three times:
// ip_adapter/attention_processor.py:187-219).  Segment 1 is IP-Adapter's image-token K/V: it gets its OWN
// softmax (two independent normalisations, attention_processor.py:482, :503, :516) — its K/V tile is staged in
// the same LDS buffers right after the text segment and folded into the same accumulator as
// (w1 / l1) * exp(s1 - m1), so decoupled cross-attention costs one extra tile, not a second kernel.
//
// Work split: block = 4 waves x 32 queries; K tile [64 keys][d] and V^T tile [d][64 keys] in LDS, shared by
// the 4 waves.  Everything is computed TRANSPOSED so that lane&31 is the query:
//   S^T[key][q]  = mfma32x32x16(A = K rows,   B = Q rows)      -> a lane holds 32 of the 64 scores of ITS query
//   O^T[d][q]   += mfma32x32x16(A = V^T rows, B = P^T)         -> P^T is consumed straight from the S^T
// registers (the MFMA C layout of S^T is a valid B-operand layout once V^T's keys are read in the matching
// permuted order: two ds_read_b64 per fragment), running max / sum / rescale are lane-local, and the only
// cross-lane traffic of the online softmax is one exchange with lane^32.
// head_dim 40 / 80 (SD-1.5) are zero-padded to 48 / 80 for QK^T (K of the MFMA) and 64 / 96 rows for PV.
#include "tg_common.h"

namespace {

constexpr int KV = 64;        // keys per tile

// LDS-DMA sources for out-of-range pieces: head-dim / key padding reads zeros; the "ones row" (below) reads 1.0
__device__ __attribute__((aligned(256))) unsigned char attn_zero_page[256];
#define TG_R8(x) x, x, x, x, x, x, x, x
__device__ __attribute__((aligned(256))) const unsigned short attn_ones_bf16[64] = {TG_R8(TG_R8(0x3F80))};
__device__ __attribute__((aligned(256))) const unsigned short attn_ones_f16[64] = {TG_R8(TG_R8(0x3C00))};
// FOLD variants: K's first padding column (d = head dim) is fed 1.0 so that the QK^T MFMA adds the query's running-max bias
__device__ __attribute__((aligned(256))) const unsigned short attn_one0_bf16[8] = {0x3F80, 0, 0, 0, 0, 0, 0, 0};
__device__ __attribute__((aligned(256))) const unsigned short attn_one0_f16[8] = {0x3C00, 0, 0, 0, 0, 0, 0, 0};
template <typename T> __device__ __forceinline__ const T* one0_page();
template <> __device__ __forceinline__ const bf16_t* one0_page<bf16_t>() { return reinterpret_cast<const bf16_t*>(attn_one0_bf16); }
template <> __device__ __forceinline__ const f16_t* one0_page<f16_t>() { return reinterpret_cast<const f16_t*>(attn_one0_f16); }
template <typename T> __device__ __forceinline__ const T* ones_page();
template <> __device__ __forceinline__ const bf16_t* ones_page<bf16_t>() { return reinterpret_cast<const bf16_t*>(attn_ones_bf16); }
template <> __device__ __forceinline__ const f16_t* ones_page<f16_t>() { return reinterpret_cast<const f16_t*>(attn_ones_f16); }

struct AttnParams {
  int heads, hd, n_q, n_qblk;
  const void* q; long q_ld, q_bs;
  const void* k0; long k0_ld, k0_bs;
  const void* vt0; long vt0_ld, vt0_bs;
  int len0;
  const void* k1; long k1_ld, k1_bs;
  const void* vt1; long vt1_ld, vt1_bs;
  int len1;
  float scale_log2;
  float w1;
  const float* w1_dev;
  void* out; long out_ld, out_bs;
  int causal;
  const float* mask; long mask_bs, mask_hs, mask_qs;   // MASK instances: additive score bias mask[b*bs + h*hs + q*qs + key] (fp32, score units)
  float inv_scale;
};

// Data path: K tile = NP panels of [64 keys][64 d] and V^T tile = [DV d][64 keys], both as 128-byte LDS rows filled by
// LDS-DMA (global_load_lds_dwordx4, no VGPR round trip, no ds_write pass) with the GEMM's XOR swizzle (16-byte slot ^
// ((row >> 1) & 7), applied on the source address and again on the fragment read).  Two stages: the DMA of tile t+1
// is in flight while tile t is multiplied, one barrier per tile.
// ONES (variants whose DV exceeds the head dim): V^T row DV-1 is fed from a page of 1.0, so the PV MFMA itself
// accumulates the softmax denominator l = sum_k p[k] in accumulator row DV-1 (rescaled with O for free) and the 32
// per-tile v_add_f32 of the row sum disappear — the loop is VALU-bound (32 v_exp_f32 at quarter rate + ~100 other
// VALU ops against 14 MFMAs per wave-tile at head dim 40), so every removed VALU instruction counts.
// FOLD (head dim = DPAD - 8, i.e. 40 -> 48: the SD-1.5 level-0 layers that dominate attention time): the loop is VALU-bound and a
// quarter of its VALU instructions were the `fma(s, scale, -max)` in front of every exp2.  Q is pre-multiplied by scale * log2(e)
// once per block, and the spare K column of the padded QK^T contraction carries 1.0 while the query's spare element carries
// -(running reference) — so the MFMA itself delivers exp2's argument and the 32 fmas per tile disappear.  The reference is a
// storage-dtype value (softmax is invariant to it); it moves lazily as before, and a move shifts the current tile's scores by the
// exact difference of the two representable values.
// MASK (own instantiations, never the hot path): `attention_mask` of the diffusers processors — an additive bias on segment 0's scores
// (`baddbmm(mask, q, k^T, beta=1, alpha=scale)`, ip_adapter/attention_processor.py:193-206), broadcast over heads and / or queries by
// zero strides.  It is added in raw-score units (mask / scale) right after the QK^T tile, so the online softmax is unchanged.
template <typename T, int DPAD, int DV, bool ONES, bool FOLD, bool PIPE, bool MASK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DV <= 64 ? 3 : (DV <= 96 ? 2 : 1)))) void attention_kernel(AttnParams p) {
  typedef typename Vec<T>::v8 V8;
  typedef typename Vec<T>::v4 V4;
  constexpr int NP = (DPAD + 63) / 64;          // K panels
  constexpr int NKS = DPAD / 16;
  constexpr int DT = DV / 32;
  constexpr int KJ = NP * 2;                    // K DMA instructions per wave per tile (8 rows x 8 slots each)
  constexpr int VJ = DV / 32;                   // V^T DMA instructions per wave per tile
  constexpr int K_ELEMS = NP * 64 * 64;
  constexpr int STAGE = K_ELEMS + DV * 64;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sbase = reinterpret_cast<T*>(smem);         // [2][ K: NP x 64 x 64 | V^T: DV x 64 ]

  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware block order (speed only): block b runs on XCD b % 8; give each XCD a contiguous chunk of the
  // (batch, head, q-block) space so that all q-blocks of one (batch, head) share K / V^T through ONE private L2.
  int lbid;
  {
    const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, xcd = blockIdx.x & 7;
    lbid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  }
  const int qblk = lbid % p.n_qblk;
  const int h = (lbid / p.n_qblk) % p.heads, b = lbid / (p.n_qblk * p.heads);
  const int HD = p.hd;
  const long qrow = (long)qblk * 128 + wave * 32 + l31;
  const bool q_ok = qrow < p.n_q;

  // Q fragments (B operand): this lane's query row, d = ks*16 + hi*8 .. +8
  V8 qf[NKS];
  {
    const T* qp = reinterpret_cast<const T*>(p.q) + (long)b * p.q_bs + qrow * p.q_ld + (long)h * HD;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int d = ks * 16 + hi * 8;
      V8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = from_f32<T>(0.f);
      if (q_ok && d < HD) v = *reinterpret_cast<const V8*>(qp + d);
      if constexpr (FOLD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = from_f32<T>(to_f32<T>(v[j]) * p.scale_log2);
      }
      qf[ks] = v;
    }
  }
  float qbias = 0.f;                       // FOLD: value of the query's spare element qf[NKS - 1][0] (lanes hi = 1) = -reference, log2 units

  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // ---- LDS-DMA tile loader
  const int lrow = lane >> 3, slot = lane & 7;
  const T* zero = reinterpret_cast<const T*>(attn_zero_page);
  auto dma = [&](const T* src, T* lds_row_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_row_base, 16, 0, 0);
  };
  auto issue = [&](const T* kb, long k_ld, const T* vb, long vt_ld, int len, int kv0, int stage) {
    T* sK = sbase + stage * STAGE;
    T* sV = sK + K_ELEMS;
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      const int q = j * 4 + wave;                       // rows [8q, 8q+8) of the panel stack
      const int prow = 8 * q + lrow;
      const int r = prow & 63, panel = prow >> 6;
      const int d0 = panel * 64 + ((slot ^ ((r >> 1) & 7)) << 3);
      const bool ok = d0 < HD && kv0 + r < len;
      const T* src = ok ? kb + (long)(kv0 + r) * k_ld + d0 : zero;
      if (FOLD && d0 == HD) src = one0_page<T>();
      dma(src, sK + q * 512);
    }
#pragma unroll
    for (int j = 0; j < VJ; ++j) {
      const int q = j * 4 + wave;
      const int d = 8 * q + lrow;
      const int c0 = kv0 + ((slot ^ ((d >> 1) & 7)) << 3);
      const T* src = (d < HD && c0 < len) ? vb + (long)d * vt_ld + c0 : zero;
      if (ONES && d == DV - 1) src = ones_page<T>();
      dma(src, sV + q * 512);
    }
  };
  // a 16-byte V^T chunk that straddles `len` brought columns >= len along (they may hold anything, and 0 * NaN = NaN
  // in the PV MFMA): zero them in LDS.  Only ragged tiles (text 77 = 64 + 13 keys, 4 image tokens) take this path.
  auto fixup_v = [&](int rem, int stage) {
    T* sV = sbase + stage * STAGE + K_ELEMS;
    if (tid < DV && tid < HD) {
      const int d = tid, ch = rem >> 3;
      T* rowp = sV + d * 64 + ((ch ^ ((d >> 1) & 7)) << 3);
      for (int c = rem & 7; c < 8; ++c) rowp[c] = from_f32<T>(0.f);
    }
  };

  const int skey = (l31 >> 1) & 7;                       // swizzle key of this lane's fragment rows
  // per-lane LDS element offsets of the fragment reads, computed ONCE (they only depend on the lane): inside the loop an
  // access is (stage base + table entry + compile-time constant) instead of re-deriving xor / shift / add per read
  // KEY PERMUTATION: MFMA row i of a 32-key score tile is fed K row perm(i) = i with bits 2 and 3 swapped.  The 32x32
  // accumulator gives lane-half `hi` the rows 8g + 4 hi + j in registers 4g + j; with the permutation registers
  // 8cc .. 8cc+7 hold the 8 CONSECUTIVE keys 16cc + 8 hi + 0..7, i.e. the P^T fragment of the PV MFMA pairs with ONE
  // 16-byte V^T slot (a single ds_read_b128 per fragment, no register shuffles) instead of two 8-byte halves of
  // neighbouring slots.
  const int prow = (l31 & 19) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
  const int pkey = (prow >> 1) & 7;
  int kofs[4], vofs[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) kofs[c] = prow * 64 + (((2 * c + hi) ^ pkey) << 3);
#pragma unroll
  for (int c = 0; c < 4; ++c) vofs[c] = l31 * 64 + (((2 * c + hi) ^ skey) << 3);
  // RAW scores of one 64-key tile for this lane's query (the softmax scale is folded into the exp2 argument by one
  // fma per element); keys >= len are masked to -inf only on a ragged tile, full tiles take no compare/select at all.
  auto scores = [&](f32x16 (&s)[2], const T* sK, int len, int kv0, bool causal) {
#pragma unroll
    for (int kvt = 0; kvt < 2; ++kvt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kvt][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        V8 kf = *reinterpret_cast<const V8*>(sK + kofs[ks & 3] + ((ks >> 2) * 4096 + kvt * 32 * 64));
        s[kvt] = mfma32(kf, qf[ks], s[kvt]);
      }
    }
    if (kv0 + KV > len || causal) {
      // ragged tile and / or causal mask (key j visible to query i iff j <= i: key 0 is always visible, so a row's running
      // max is finite from the first tile on and fully masked later tiles contribute exp2(-inf) = 0)
      const int qmax = causal ? (int)qrow : 0x7fffffff;
#pragma unroll
      for (int kvt = 0; kvt < 2; ++kvt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + kvt * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);     // key held by register r (permuted rows)
          if (kv >= len || kv > qmax) s[kvt][r] = -INFINITY;
        }
    }
  };

  auto tile_max = [&](const f32x16 (&s)[2]) {
    float mx = -INFINITY;
#pragma unroll
    for (int kvt = 0; kvt < 2; ++kvt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kvt][r]);
    // the other 32-key half of the row lives in lane ^ 32: v_permlane32_swap exchanges the upper 32 lanes of one copy
    // with the lower 32 of the other (no LDS round trip, unlike the ds_bpermute behind __shfl_xor)
    // (inline asm: with the builtin hipcc drops the second result and the max with it; s_nop covers the VALU-write ->
    // permlane-read hazard the assembler does not see)
    float a = mx, b = mx;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
  };

  // O^T += V^T * P^T for one tile; P^T comes straight from the score registers: chunk c covers keys
  // kvt*32 + 16*cc + 8*hi + 0..7 = 16-byte slot 4*kvt + 2*cc + hi of the V^T row
  auto pv = [&](const f32x16 (&s)[2], const T* sV) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int kvt = c >> 1, cc = c & 1;
      V8 pf;
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = from_f32<T>(s[kvt][8 * cc + j]);
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const V8 vf = *reinterpret_cast<const V8*>(sV + vofs[c] + t * 32 * 64);
        o[t] = mfma32(vf, pf, o[t]);
      }
    }
  };

  const T* kb0 = reinterpret_cast<const T*>(p.k0) + (long)b * p.k0_bs + (long)h * HD;
  const T* vb0 = reinterpret_cast<const T*>(p.vt0) + (long)b * p.vt0_bs + (long)h * HD * p.vt0_ld;
  const T* kb1 = reinterpret_cast<const T*>(p.k1) + (long)b * p.k1_bs + (long)h * HD;
  const T* vb1 = reinterpret_cast<const T*>(p.vt1) + (long)b * p.vt1_bs + (long)h * HD * p.vt1_ld;

  // ---------------- segment 0: online softmax over len0 keys
  const int nt0 = (p.len0 + KV - 1) / KV;
  issue(kb0, p.k0_ld, vb0, p.vt0_ld, p.len0, 0, 0);
  for (int t = 0; t < nt0; ++t) {
    const int st = t & 1, kv0 = t * KV;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kv0 + KV > p.len0 && ((p.len0 - kv0) & 7)) {
      fixup_v(p.len0 - kv0, st);
      __syncthreads();
    }
    if (t + 1 < nt0) issue(kb0, p.k0_ld, vb0, p.vt0_ld, p.len0, kv0 + KV, st ^ 1);
    else if (p.len1 > 0) issue(kb1, p.k1_ld, vb1, p.vt1_ld, p.len1, 0, st ^ 1);
    const T* sK = sbase + st * STAGE;
    f32x16 s[2];
    // PIPE (DV = 64 variants: head dims 40 and 64), full tiles only: MFMA and VALU work of ONE wave interleaved in program order.  The
    // chain QK^T -> max -> exp2 -> convert -> PV is dependent end to end inside a wave, and the rocprofv3 trace of the SD-1.5 step
    // (5 level-0 self-attention launches of 499 us = 1096 cycles per wave-tile and SIMD against ~450 of VALU + ~450 of matrix pipe)
    // says the three waves of a SIMD do not hide each other's phases.  So: the second 32-key half's QK^T MFMAs run with the max over
    // the first half in their shadow, and the PV MFMAs of the first half run with the exp2s of the second half between them (an MFMA
    // executes asynchronously; the wave keeps issuing independent VALU instructions behind it).  Same arithmetic, same order of
    // every accumulation: bit-identical to the straight-line path (which ragged / causal tiles keep).  Measured (MI355X, same box,
    // TG_ATTN_FLAGS 1 / 0): level-0 self-attention in isolation 660 -> 634 us and 719 -> 698 us on two boxes (-3 .. -4 %), the SD-1.5
    // bench +0.2 .. +0.6 % — at 152 instead of 125 VGPRs (three resident waves per SIMD instead of four).  What it says about the
    // hardware: a wave-tile costs about the SUM of its VALU and matrix-pipe time whatever the issue order, so the remaining levers
    // are fewer instructions, not more overlap.
    // PIPE is a TEMPLATE parameter (its own instantiation, dev switch TG_ATTN_PIPE=1): compiled into the default kernel behind a
    // run-time flag (round 3) it took the d = 40 kernel from 125 VGPRs / 0 scratch to 168 VGPRs / 6 spills with scratch reloads
    // inside the tile loop (+30 % per launch) — the default instantiation carries none of it.
    static_assert(!PIPE || (DV == 64 && NKS <= 3), "PIPE: head dim 40 only");
    const bool piped = PIPE && !(kv0 + KV > p.len0 || p.causal != 0);
    float tm;
    if (piped) {
      V8 kf0[NKS], kf1[NKS];
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        kf0[ks] = *reinterpret_cast<const V8*>(sK + kofs[ks & 3] + (ks >> 2) * 4096);
        kf1[ks] = *reinterpret_cast<const V8*>(sK + kofs[ks & 3] + ((ks >> 2) * 4096 + 32 * 64));
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[0][r] = 0.f; s[1][r] = 0.f; }
      float mx = -INFINITY;
      // S0 chain first, the S1 chain with the max over S0 in its shadow
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) s[0] = mfma32(kf0[ks], qf[ks], s[0]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        s[1] = mfma32(kf1[ks], qf[ks], s[1]);
#pragma unroll
        for (int r = ks * 16 / NKS; r < (ks + 1) * 16 / NKS; ++r) mx = fmaxf(mx, s[0][r]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
      float a = mx, b2 = mx;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b2));
      tm = fmaxf(a, b2);
    } else {
      scores(s, sK, p.len0, kv0, p.causal != 0);
      if constexpr (MASK) {
        const float* mp = p.mask + (long)b * p.mask_bs + (long)h * p.mask_hs + (q_ok ? qrow : 0) * p.mask_qs;
#pragma unroll
        for (int kvt = 0; kvt < 2; ++kvt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kv = kv0 + kvt * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
            if (kv < p.len0) s[kvt][r] += mp[kv] * p.inv_scale;
          }
      }
      tm = tile_max(s);
    }
    // m_run is kept in RAW score units; exp2 arguments are formed as fma(s, c, -m*c) with c = scale * log2(e) > 0.
    // v_exp_f32 directly (__builtin_amdgcn_exp2f): results stay far inside the fp32 range, exp2(-inf) = 0 — none of
    // exp2f()'s denormal-range rescaling (v_ldexp + compares + selects per element) is needed.
    // LAZY RUNNING MAX: softmax is invariant to the reference point, so m_run only moves (and O is only rescaled: 32
    // accumulator multiplies per tile) when some row's tile max exceeds it by more than LAZY_LOG2 in exp2 units; until
    // then probabilities are formed against the stale reference and are at most 2^LAZY_LOG2 (bf16 P and the fp32
    // accumulators have the range).  With a strict "any row has a new max" test the rescale ran on most tiles: over 32
    // rows a new maximum keeps turning up somewhere.
    constexpr float LAZY_LOG2 = 8.f;
    float ps = 0.f;
    float mc = 0.f;                                  // generic path: the reference in exp2 units
    if constexpr (FOLD) {
      // s already holds exp2's argument relative to the current reference (-qbias)
      if (t == 0 || __any(tm > LAZY_LOG2)) {
        const float sh = t == 0 ? tm : fmaxf(tm, 0.f);
        const float nb = to_f32<T>(from_f32<T>(qbias - sh));            // the new bias must be a storage-dtype value
        const float delta = nb - qbias;                                 // exact: both are representable
        const float alpha = t == 0 ? 1.0f : __builtin_amdgcn_exp2f(delta);   // first tile: O and l are still zero
#pragma unroll
        for (int t2 = 0; t2 < DT; ++t2)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[t2][r] *= alpha;
        if (!ONES) l_run *= alpha;
#pragma unroll
        for (int kvt = 0; kvt < 2; ++kvt)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kvt][r] += delta;
        qbias = nb;
        if (hi) qf[NKS - 1][0] = from_f32<T>(nb);
      }
    } else {
      if (__any(tm * p.scale_log2 > m_run * p.scale_log2 + LAZY_LOG2)) {
        const float m_new = fmaxf(m_run, tm);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2);
#pragma unroll
        for (int t2 = 0; t2 < DT; ++t2)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[t2][r] *= alpha;
        if (!ONES) l_run *= alpha;
        m_run = m_new;
      }
      mc = m_run * p.scale_log2;
    }
    // exp2 of one score register: FOLD scores are exp2's argument already
    auto ex = [&](float x) { return FOLD ? __builtin_amdgcn_exp2f(x) : __builtin_amdgcn_exp2f(__builtin_fmaf(x, p.scale_log2, -mc)); };
    const T* sV = sK + K_ELEMS;
    if (piped) {
      V8 vf[2][DT];                                  // V^T fragments of the first half's two chunks: requested before its exp2s
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t2 = 0; t2 < DT; ++t2) vf[c][t2] = *reinterpret_cast<const V8*>(sV + vofs[c] + t2 * 32 * 64);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = ex(s[0][r]);
        s[0][r] = e;
        if (!ONES) ps += e;
      }
      __builtin_amdgcn_sched_barrier(0);
      constexpr int EPM = 16 / (2 * DT);             // exp2s of the second half behind every PV MFMA of the first
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        V8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = from_f32<T>(s[0][8 * c + j]);
#pragma unroll
        for (int t2 = 0; t2 < DT; ++t2) {
          o[t2] = mfma32(vf[c][t2], pf, o[t2]);
#pragma unroll
          for (int r = (c * DT + t2) * EPM; r < (c * DT + t2 + 1) * EPM; ++r) {
            const float e = ex(s[1][r]);
            s[1][r] = e;
            if (!ONES) ps += e;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int c = 2; c < 4; ++c) {
        V8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = from_f32<T>(s[1][8 * (c & 1) + j]);
#pragma unroll
        for (int t2 = 0; t2 < DT; ++t2) {
          const V8 vf2 = *reinterpret_cast<const V8*>(sV + vofs[c] + t2 * 32 * 64);
          o[t2] = mfma32(vf2, pf, o[t2]);
        }
      }
      if (!ONES) l_run += ps;
    } else {
#pragma unroll
      for (int kvt = 0; kvt < 2; ++kvt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = ex(s[kvt][r]);
          s[kvt][r] = e;
          if (!ONES) ps += e;
        }
      if (!ONES) l_run += ps;
      pv(s, sV);
    }
  }
  {
    float l_tot;
    if (ONES) {
      // accumulator row DV-1 = tile DT-1, register 15 of the lanes with hi = 1
      const float c = hi ? o[DT - 1][15] : 0.f;
      l_tot = c + __shfl_xor(c, 32, 64);
    } else {
      l_tot = l_run + __shfl_xor(l_run, 32, 64);
    }
    const float inv = 1.f / l_tot;
#pragma unroll
    for (int t2 = 0; t2 < DT; ++t2)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t2][r] *= inv;
  }

  // ---------------- segment 1 (single tile, own softmax), folded in with weight w1 / l1
  if (p.len1 > 0) {
    const int st = nt0 & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (p.len1 < KV && (p.len1 & 7)) {
      fixup_v(p.len1, st);
      __syncthreads();
    }
    const T* sK = sbase + st * STAGE;
    f32x16 s[2];
    if constexpr (FOLD) {
      if (hi) qf[NKS - 1][0] = from_f32<T>(0.f);            // segment 1 has its own softmax: no carried reference
    }
    scores(s, sK, p.len1, 0, false);
    const float sc1 = FOLD ? 1.0f : p.scale_log2;           // FOLD: the scores are already in log2 units
    const float mc1 = tile_max(s) * sc1;
    float ps = 0.f;
#pragma unroll
    for (int kvt = 0; kvt < 2; ++kvt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kvt][r], sc1, -mc1));
        s[kvt][r] = e;
        ps += e;
      }
    const float l1 = ps + __shfl_xor(ps, 32, 64);
    const float f = (p.w1_dev != nullptr ? *p.w1_dev : p.w1) / l1;      // device scalar: the IP scale may change between replays of a captured graph
#pragma unroll
    for (int kvt = 0; kvt < 2; ++kvt)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kvt][r] *= f;
    pv(s, sK + K_ELEMS);
  }

  // ---------------- store: O^T regs -> out[b, q, h*HD + d], 4 consecutive d per 8-byte store
  if (q_ok) {
    T* op = reinterpret_cast<T*>(p.out) + (long)b * p.out_bs + qrow * p.out_ld + (long)h * HD;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = t * 32 + 8 * g + 4 * hi;
        if (d < HD) {
          V4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = from_f32<T>(o[t][4 * g + j]);
          *reinterpret_cast<V4*>(op + d) = v;
        }
      }
  }
}

template <typename T, int DPAD, int DV, bool ONES, bool FOLD = false, bool PIPE = false, bool MASK = false>
int launch_attn(const tg_attn_desc* d, const AttnParams& p, hipStream_t st) {
  constexpr int NP = (DPAD + 63) / 64;
  const size_t lds = (size_t)2 * (NP * 64 * 64 + DV * 64) * sizeof(T);
  AttnParams pp = p;
  pp.n_qblk = (d->n_q + 127) / 128;
  dim3 grid((unsigned)(pp.n_qblk * d->heads * d->batch));
  auto k = attention_kernel<T, DPAD, DV, ONES, FOLD, PIPE, MASK>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr;
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, pp);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

template <typename T>
int dispatch_attn(const tg_attn_desc* d, const AttnParams& p, hipStream_t st) {
  // ONES variants need a spare V^T row (DV > head dim): 40 -> (48, 64), 80 -> (80, 96), <= 16 -> (16, 32)
  const int hd = d->head_dim;
  if (p.mask != nullptr) {
    if (hd <= 16) return launch_attn<T, 16, 32, true, false, false, true>(d, p, st);
    if (hd <= 32) return launch_attn<T, 32, 32, false, false, false, true>(d, p, st);
    if (hd <= 48) return launch_attn<T, 48, 64, true, false, false, true>(d, p, st);
    if (hd <= 64) return launch_attn<T, 64, 64, false, false, false, true>(d, p, st);
    if (hd <= 80) return launch_attn<T, 80, 96, true, false, false, true>(d, p, st);
    if (hd <= 96) return launch_attn<T, 96, 96, false, false, false, true>(d, p, st);
    if (hd <= 128) return launch_attn<T, 128, 128, false, false, false, true>(d, p, st);
    return launch_attn<T, 160, 160, false, false, false, true>(d, p, st);
  }
  if (hd <= 16) return launch_attn<T, 16, 32, true>(d, p, st);
  if (hd <= 32) return launch_attn<T, 32, 32, false>(d, p, st);
  if (hd == 40 && !getenv("TG_ATTN_NOFOLD")) {                                                       // bias folded into QK^T (dev switch for A/B)
    static const bool pipe = getenv("TG_ATTN_PIPE") != nullptr;                                       // dev A/B: the intra-wave interleave instantiation
    return pipe ? launch_attn<T, 48, 64, true, true, true>(d, p, st) : launch_attn<T, 48, 64, true, true>(d, p, st);
  }
  if (hd <= 48) return launch_attn<T, 48, 64, true>(d, p, st);
  if (hd <= 64) return launch_attn<T, 64, 64, false>(d, p, st);
  if (hd <= 80) return launch_attn<T, 80, 96, true>(d, p, st);
  if (hd <= 96) return launch_attn<T, 96, 96, false>(d, p, st);
  if (hd <= 128) return launch_attn<T, 128, 128, false>(d, p, st);
  return launch_attn<T, 160, 160, false>(d, p, st);
}

// ---- attention-probability export (save_attn_to_dict side channel): one wave per query row --------------
template <typename T>
__global__ __launch_bounds__(256) void attn_probs_kernel(int b0, int heads, int hd, int n_q, const T* q, long q_ld, long q_bs,
                                                         const T* k, long k_ld, long k_bs, int len, float scale,
                                                         const int* tokens, int n_tokens, float* probs) {
  typedef typename Vec<T>::v8 V8;
  const int lane = threadIdx.x & 63;
  const long qi = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int h = blockIdx.y, bb = blockIdx.z, b = b0 + bb;
  if (qi >= n_q) return;
  const T* qp = q + (long)b * q_bs + qi * q_ld + (long)h * hd;
  const T* kp = k + (long)b * k_bs + (long)h * hd;
  constexpr int MAXJ = 4;  // keys per lane (len <= 256)
  float sc[MAXJ];
#pragma unroll
  for (int jj = 0; jj < MAXJ; ++jj) {
    const int j = lane + 64 * jj;
    float acc = 0.f;
    if (j < len) {
      for (int d = 0; d < hd; d += 8) {
        V8 a = *reinterpret_cast<const V8*>(qp + d);
        V8 c = *reinterpret_cast<const V8*>(kp + (long)j * k_ld + d);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += to_f32<T>(a[e]) * to_f32<T>(c[e]);
      }
    }
    sc[jj] = j < len ? acc * scale : -INFINITY;
  }
  float mx = -INFINITY;
#pragma unroll
  for (int jj = 0; jj < MAXJ; ++jj) mx = fmaxf(mx, sc[jj]);
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int jj = 0; jj < MAXJ; ++jj) { sc[jj] = __expf(sc[jj] - mx); sum += sc[jj]; }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  const int nt = tokens ? n_tokens : len;
  float* op = probs + (((long)bb * heads + h) * n_q + qi) * nt;
  if (!tokens) {
#pragma unroll
    for (int jj = 0; jj < MAXJ; ++jj) {
      const int j = lane + 64 * jj;
      if (j < len) op[j] = sc[jj] * inv;
    }
  } else {
    for (int t = 0; t < n_tokens; ++t) {
      const int j = tokens[t];
      const int owner = j & 63, slot = j >> 6;
      float v = 0.f;
#pragma unroll
      for (int jj = 0; jj < MAXJ; ++jj)
        if (jj == slot) v = sc[jj];
      v = __shfl(v, owner, 64);
      if (lane == 0) op[t] = v * inv;
    }
  }
}

}  // namespace

extern "C" int tg_attention(const tg_attn_desc* d, void* stream) {
  TG_CHECK(d != nullptr, TG_ERR_ARG, "tg_attention: null descriptor");
  TG_CHECK(d->dtype == TG_BF16 || d->dtype == TG_F16, TG_ERR_ARG, "tg_attention: bad dtype");
  TG_CHECK(d->batch > 0 && d->heads > 0 && d->n_q > 0 && d->len0 > 0, TG_ERR_ARG, "tg_attention: empty problem");
  TG_CHECK(d->head_dim > 0 && d->head_dim % 8 == 0 && d->head_dim <= 160, TG_ERR_ARG,
           "tg_attention: head_dim %d unsupported (multiple of 8, <= 160)", d->head_dim);
  TG_CHECK(d->q && d->k0 && d->vt0 && d->out, TG_ERR_ARG, "tg_attention: null q/k0/vt0/out");
  TG_CHECK(d->q_ld % 8 == 0 && d->k0_ld % 8 == 0 && d->vt0_ld % 8 == 0 && d->out_ld % 4 == 0, TG_ERR_ARG,
           "tg_attention: pitches must keep 16-byte alignment");
  TG_CHECK(d->len1 >= 0 && d->len1 <= KV, TG_ERR_ARG, "tg_attention: len1 (%d) must be <= 64", d->len1);
  if (d->len1 > 0) TG_CHECK(d->k1 && d->vt1 && d->k1_ld % 8 == 0 && d->vt1_ld % 8 == 0, TG_ERR_ARG, "tg_attention: bad segment 1");
  AttnParams p{};
  p.heads = d->heads; p.hd = d->head_dim; p.n_q = d->n_q;
  p.q = d->q; p.q_ld = d->q_ld; p.q_bs = d->q_bs;
  p.k0 = d->k0; p.k0_ld = d->k0_ld; p.k0_bs = d->k0_bs; p.vt0 = d->vt0; p.vt0_ld = d->vt0_ld; p.vt0_bs = d->vt0_bs; p.len0 = d->len0;
  p.k1 = d->k1; p.k1_ld = d->k1_ld; p.k1_bs = d->k1_bs; p.vt1 = d->vt1; p.vt1_ld = d->vt1_ld; p.vt1_bs = d->vt1_bs; p.len1 = d->len1;
  p.scale_log2 = d->scale * 1.4426950408889634f;
  p.w1 = d->w1;
  p.w1_dev = d->w1_dev;
  p.out = d->out; p.out_ld = d->out_ld; p.out_bs = d->out_bs;
  p.causal = d->causal;
  if (d->mask != nullptr) {
    TG_CHECK(d->scale > 0.f, TG_ERR_ARG, "tg_attention: an additive mask needs scale > 0");
    p.mask = d->mask; p.mask_bs = d->mask_bs; p.mask_hs = d->mask_hs; p.mask_qs = d->mask_qs;
    p.inv_scale = 1.0f / d->scale;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == TG_BF16) return dispatch_attn<bf16_t>(d, p, st);
  return dispatch_attn<f16_t>(d, p, st);
}

extern "C" int tg_attn_probs(int32_t dtype, int32_t batch, int32_t b0, int32_t heads, int32_t head_dim, int32_t n_q,
                             const void* q, int64_t q_ld, int64_t q_bs, const void* k, int64_t k_ld, int64_t k_bs,
                             int32_t len, float scale, const int32_t* tokens, int32_t n_tokens, float* probs, void* stream) {
  TG_CHECK(dtype == TG_BF16 || dtype == TG_F16, TG_ERR_ARG, "tg_attn_probs: bad dtype");
  TG_CHECK(q && k && probs && batch > b0 && b0 >= 0 && heads > 0 && n_q > 0, TG_ERR_ARG, "tg_attn_probs: bad args");
  TG_CHECK(len > 0 && len <= 256 && head_dim % 8 == 0, TG_ERR_ARG, "tg_attn_probs: len (%d) must be in 1..256", len);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((unsigned)((n_q + 3) / 4), (unsigned)heads, (unsigned)(batch - b0));
  if (dtype == TG_BF16)
    hipLaunchKernelGGL(attn_probs_kernel<bf16_t>, grid, dim3(256), 0, st, b0, heads, head_dim, n_q, (const bf16_t*)q, q_ld, q_bs,
                       (const bf16_t*)k, k_ld, k_bs, len, scale, tokens, n_tokens, probs);
  else
    hipLaunchKernelGGL(attn_probs_kernel<f16_t>, grid, dim3(256), 0, st, b0, heads, head_dim, n_q, (const f16_t*)q, q_ld, q_bs,
                       (const f16_t*)k, k_ld, k_bs, len, scale, tokens, n_tokens, probs);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

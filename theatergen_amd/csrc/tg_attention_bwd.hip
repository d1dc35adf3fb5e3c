// Recompute-based (flash-style) reverse pass of self-attention, head dim <= 64 (round 5; tg_attention_bwd).
//
// The guidance reverse pass (models/pipelines.py:62-128: `torch.autograd.grad(loss, latents)` through the whole UNet) needs, per self-attention layer,
//     dQ = dS K,  dK = dS^T Q,  dV = P^T dO,      P = softmax(s Q K^T),  dS = s P o (dO V^T - D),  D[q] = sum_k P[q, k] (dO V^T)[q, k]
// (`Attention` / AttnProcessor, ip_adapter/attention_processor.py:113-219; autograd's softmax / bmm backward).  Rounds 3-4 materialised P and dS per
// (batch item, head) — nine launches each, N x N matrices through HBM (9216 x 9216 x 2 B = 170 MB per head at SD-2.1's first level).  Here nothing
// N x N exists: three launches of ONE kernel template per layer, all (item, head) pairs in the grid, scores recomputed from Q / K tiles:
//   MODE 2 (statistics):  per query row  lse2 = log2 sum_k 2^(c s_qk)  and  D  — online over the key tiles (running max, sum, sum of p * dP);
//   MODE 0 (dQ):          a workgroup owns 128 queries (Q, dO as MFMA B operands in registers), streams K / V / K^T tiles:
//                           S^T = K Q^T,  dP^T = V dO^T,  P^T = 2^(c S^T - lse2),  dS^T = s P^T o (dP^T - D),  dQ^T += K^T dS^T;
//   MODE 1 (dK, dV):      the same with the roles swapped — a workgroup owns 128 KEYS (K, V in registers), streams Q / dO / Q^T / dO^T tiles and the
//                           tile's (lse2, D) pairs:  S = Q K^T,  dP = dO V^T,  P,  dS,   dK^T += Q^T dS,   dV^T += dO^T P.
// Data path = the forward kernel's (tg_attention.hip): 64-row tiles as 128-byte LDS rows filled by LDS-DMA with the XOR swizzle, two stages, one
// barrier per tile; score tiles with the streamed rows on the ACCUMULATOR rows in the bit-2/3-swapped order, so that 8 consecutive registers are 8
// consecutive streamed rows = the B fragment of the following product as they stand (P and dS are rounded to the storage dtype there — what the
// materialised path stored).  Deterministic: no atomics, every output element is written by one lane.
#include "tg_common.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int attn_bwd_zero_page[4] = {0u, 0u, 0u, 0u};

struct AttnBwdParams {
  int heads, hd, n_r, n_s, n_rblk;
  int n_z;                                               // columns of the transposed streamed tensors that may be read (n_s rounded up to 8: zero padding)
  const float* extra; long e_ld;                         // MODE 0 / 2, optional: d loss / d P added to dP, fp32 [batch][heads][n_r][e_ld >= n_s] (cross-attention guidance term)
  float ds_scale;                                        // dS = ds_scale * P o (dP - D)   (softmax scale x the segment's output weight)
  const void* r1; const void* r2; long r_ld, r_bs;       // register-side rows [n_r][...]: MODE 0 / 2: Q, dO;  MODE 1: K, V
  const void* s1; const void* s2; long s_ld, s_bs;       // streamed rows [n_s][...]:      MODE 0 / 2: K, V;   MODE 1: Q, dO
  const void* z1; const void* z2; long z_ld, z_bs;       // streamed tensors transposed [inner][n_s]: MODE 0: K^T;  MODE 1: Q^T, dO^T
  float* stats;                                          // [batch][heads][n_q][2] = (lse2, D)
  void* out1; void* out2; long o_ld, o_bs;               // MODE 0: dQ;  MODE 1: dK, dV   ([rows][inner] like the inputs)
  float scale, scale_log2;
};

template <typename T, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn_bwd_kernel(AttnBwdParams p) {
  typedef typename Vec<T>::v8 V8;
  typedef typename Vec<T>::v4 V4;
  constexpr int NKS = 4, DT = 2, KV = 64, PANEL = 64 * 64, STAGE = 4 * PANEL + 512;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sbase = reinterpret_cast<T*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lbid;
  {
    const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, xcd = blockIdx.x & 7;
    lbid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  }
  const int rblk = lbid % p.n_rblk;
  const int h = (lbid / p.n_rblk) % p.heads, b = lbid / (p.n_rblk * p.heads);
  const int HD = p.hd;
  const long rrow = (long)rblk * 128 + wave * 32 + l31;
  const bool r_ok = rrow < p.n_r;

  // register-side fragments (B operands): this lane's row, d = ks*16 + hi*8 .. +8
  V8 r1f[NKS], r2f[NKS];
  {
    const T* p1 = reinterpret_cast<const T*>(p.r1) + (long)b * p.r_bs + rrow * p.r_ld + (long)h * HD;
    const T* p2 = reinterpret_cast<const T*>(p.r2) + (long)b * p.r_bs + rrow * p.r_ld + (long)h * HD;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int d = ks * 16 + hi * 8;
      V8 z;
#pragma unroll
      for (int j = 0; j < 8; ++j) z[j] = from_f32<T>(0.f);
      r1f[ks] = z; r2f[ks] = z;
      if (r_ok && d < HD) { r1f[ks] = *reinterpret_cast<const V8*>(p1 + d); r2f[ks] = *reinterpret_cast<const V8*>(p2 + d); }
    }
  }
  float* statb = p.stats + ((long)b * p.heads + h) * (MODE == 1 ? p.n_s : p.n_r) * 2;
  float lse_l = 0.f, d_l = 0.f;
  if (MODE == 0 && r_ok) { lse_l = statb[2 * rrow]; d_l = statb[2 * rrow + 1]; }

  f32x16 o1[DT], o2[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { o1[t][r] = 0.f; o2[t][r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f, num = 0.f;         // MODE 2 (per lane half: its 32 of every tile's 64 keys)

  // ---- LDS-DMA tile loader (forward kernel's scheme: instruction q covers rows [8q, 8q + 8), lane -> (row 8q + lane / 8, 16-byte slot lane % 8))
  const int lrow = lane >> 3, slot = lane & 7;
  const T* zero = reinterpret_cast<const T*>(attn_bwd_zero_page);
  auto dma = [&](const T* src, T* lds_row_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_row_base, 16, 0, 0);
  };
  const T* s1b = reinterpret_cast<const T*>(p.s1) + (long)b * p.s_bs + (long)h * HD;
  const T* s2b = reinterpret_cast<const T*>(p.s2) + (long)b * p.s_bs + (long)h * HD;
  const T* z1b = reinterpret_cast<const T*>(p.z1) + (long)b * p.z_bs + (long)h * HD * p.z_ld;
  const T* z2b = reinterpret_cast<const T*>(p.z2) + (long)b * p.z_bs + (long)h * HD * p.z_ld;
  auto issue = [&](int s0, int stage) {
    T* st = sbase + stage * STAGE;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = j * 4 + wave;
      const int row = 8 * q + lrow;
      const int d0 = (slot ^ ((row >> 1) & 7)) << 3;
      const bool ok = d0 < HD && s0 + row < p.n_s;
      dma(ok ? s1b + (long)(s0 + row) * p.s_ld + d0 : zero, st + q * 512);
      dma(ok ? s2b + (long)(s0 + row) * p.s_ld + d0 : zero, st + PANEL + q * 512);
    }
    if constexpr (MODE != 2) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int q = j * 4 + wave;
        const int d = 8 * q + lrow;
        const int c0 = s0 + ((slot ^ ((d >> 1) & 7)) << 3);
        const bool ok = d < HD && c0 < p.n_z;                // n_z % 8 == 0: a 16-byte chunk is wholly inside or wholly outside (columns >= n_s: zero padding)
        dma(ok ? z1b + (long)d * p.z_ld + c0 : zero, st + 2 * PANEL + q * 512);
        if constexpr (MODE == 1) dma(ok ? z2b + (long)d * p.z_ld + c0 : zero, st + 3 * PANEL + q * 512);
      }
    }
    if constexpr (MODE == 1) {
      if (wave == 0) {                                       // the tile's 64 (lse2, D) pairs: 512 bytes = lanes 0 .. 31
        const int idx = s0 + 2 * lane;
        const bool ok = lane < 32 && idx < p.n_s;
        dma(ok ? reinterpret_cast<const T*>(statb + 2 * idx) : zero, st + 4 * PANEL);
      }
    }
  };

  const int skey = (l31 >> 1) & 7;
  const int prow = (l31 & 19) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);      // streamed-row permutation (tg_attention.hip: KEY PERMUTATION)
  const int pkey = (prow >> 1) & 7;
  int kofs[4], vofs[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) kofs[c] = prow * 64 + (((2 * c + hi) ^ pkey) << 3);
#pragma unroll
  for (int c = 0; c < 4; ++c) vofs[c] = l31 * 64 + (((2 * c + hi) ^ skey) << 3);

  const int nt = (p.n_s + KV - 1) / KV;
  issue(0, 0);
  for (int t = 0; t < nt; ++t) {
    const int stg = t & 1, s0 = t * KV;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 1 < nt) issue(s0 + KV, stg ^ 1);
    const T* sP1 = sbase + stg * STAGE;
    const T* sP2 = sP1 + PANEL;
    // the two score-shaped products: rows = streamed items (permuted), columns = this lane's register-side row
    f32x16 s1[2], s2[2];
#pragma unroll
    for (int kvt = 0; kvt < 2; ++kvt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { s1[kvt][r] = 0.f; s2[kvt][r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const V8 f1 = *reinterpret_cast<const V8*>(sP1 + kofs[ks] + kvt * 32 * 64);
        const V8 f2 = *reinterpret_cast<const V8*>(sP2 + kofs[ks] + kvt * 32 * 64);
        s1[kvt] = mfma32(f1, r1f[ks], s1[kvt]);
        s2[kvt] = mfma32(f2, r2f[ks], s2[kvt]);
      }
    }
    const bool ragged = s0 + KV > p.n_s;
    if constexpr (MODE != 1) {
      if (p.extra != nullptr && r_ok) {
        // the guidance loss reads the probabilities themselves: d loss / d P joins dP (same (query, key) element; fp32 rows of this lane's query)
        const float* ex = p.extra + (((long)b * p.heads + h) * p.n_r + rrow) * p.e_ld;
#pragma unroll
        for (int kvt = 0; kvt < 2; ++kvt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int sidx = s0 + kvt * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
            if (sidx < p.n_s) s2[kvt][r] += ex[sidx];
          }
      }
    }
    if constexpr (MODE == 2) {
      // online statistics over this lane half's keys of the tile (register r of tile kvt = streamed row kvt*32 + 16 (r >> 3) + 8 hi + (r & 7))
      float tm = -INFINITY;
#pragma unroll
      for (int kvt = 0; kvt < 2; ++kvt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float x = s1[kvt][r] * p.scale_log2;
          if (ragged && s0 + kvt * 32 + 16 * (r >> 3) + 8 * hi + (r & 7) >= p.n_s) x = -INFINITY;
          s1[kvt][r] = x;
          tm = fmaxf(tm, x);
        }
      const float m_new = fmaxf(m_run, tm);
      const float mref = m_new == -INFINITY ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f(m_run - mref);          // first tile: exp2(-inf) = 0
      float ls = 0.f, ns = 0.f;
#pragma unroll
      for (int kvt = 0; kvt < 2; ++kvt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pe = __builtin_amdgcn_exp2f(s1[kvt][r] - mref);
          ls += pe;
          ns = __builtin_fmaf(pe, s2[kvt][r], ns);
        }
      l_run = __builtin_fmaf(l_run, alpha, ls);
      num = __builtin_fmaf(num, alpha, ns);
      m_run = m_new;
    } else {
      const float* sst = reinterpret_cast<const float*>(sP1 + 4 * PANEL);
      const T* sZ1 = sP1 + 2 * PANEL;
      const T* sZ2 = sP1 + 3 * PANEL;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int kvt = c >> 1, cc = c & 1;
        float lse8[8], d8[8];
        if constexpr (MODE == 1) {
          const float* sp = sst + 2 * (kvt * 32 + 16 * cc + 8 * hi);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(sp + 4 * q);
            lse8[2 * q] = v[0]; d8[2 * q] = v[1]; lse8[2 * q + 1] = v[2]; d8[2 * q + 1] = v[3];
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) { lse8[j] = lse_l; d8[j] = d_l; }
        }
        V8 pf, dsf;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = 8 * cc + j;
          float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[kvt][r], p.scale_log2, -lse8[j]));
          if (ragged && s0 + kvt * 32 + 16 * cc + 8 * hi + j >= p.n_s) pe = 0.f;
          pf[j] = from_f32<T>(pe);
          dsf[j] = from_f32<T>(pe * (s2[kvt][r] - d8[j]) * p.ds_scale);
        }
#pragma unroll
        for (int t2 = 0; t2 < DT; ++t2) {
          const V8 zf = *reinterpret_cast<const V8*>(sZ1 + vofs[c] + t2 * 32 * 64);
          o1[t2] = mfma32(zf, dsf, o1[t2]);
          if constexpr (MODE == 1) {
            const V8 zg = *reinterpret_cast<const V8*>(sZ2 + vofs[c] + t2 * 32 * 64);
            o2[t2] = mfma32(zg, pf, o2[t2]);
          }
        }
      }
    }
  }

  if constexpr (MODE == 2) {
    // merge the two lane halves' partial statistics of the row, then lse2 = m + log2(l), D = num / l
    const float m_o = __shfl_xor(m_run, 32, 64), l_o = __shfl_xor(l_run, 32, 64), n_o = __shfl_xor(num, 32, 64);
    const float mm = fmaxf(m_run, m_o);
    const float mref = mm == -INFINITY ? 0.f : mm;
    const float a = __builtin_amdgcn_exp2f(m_run - mref), a_o = __builtin_amdgcn_exp2f(m_o - mref);
    const float l_tot = l_run * a + l_o * a_o, n_tot = num * a + n_o * a_o;
    if (r_ok && hi == 0) {
      statb[2 * rrow] = mref + __builtin_amdgcn_logf(l_tot);             // v_log_f32 = log2
      statb[2 * rrow + 1] = n_tot / l_tot;
    }
    return;
  }
  // ---- store: accumulator rows d -> out[b, row, h*HD + d], 4 consecutive d per 8-byte store (forward kernel's store)
  if (r_ok) {
    T* op1 = reinterpret_cast<T*>(p.out1) + (long)b * p.o_bs + rrow * p.o_ld + (long)h * HD;
    T* op2 = MODE == 1 ? reinterpret_cast<T*>(p.out2) + (long)b * p.o_bs + rrow * p.o_ld + (long)h * HD : nullptr;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = t * 32 + 8 * g + 4 * hi;
        if (d < HD) {
          V4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = from_f32<T>(o1[t][4 * g + j]);
          *reinterpret_cast<V4*>(op1 + d) = v;
          if constexpr (MODE == 1) {
            V4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = from_f32<T>(o2[t][4 * g + j]);
            *reinterpret_cast<V4*>(op2 + d) = w;
          }
        }
      }
  }
}

template <typename T, int MODE>
void launch_bwd(const AttnBwdParams& p, int batch, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * (4 * 64 * 64 + 512) * sizeof(T);
  auto k = attn_bwd_kernel<T, MODE>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr;
  hipLaunchKernelGGL(k, dim3((unsigned)(p.n_rblk * p.heads * batch)), dim3(256), lds, st, p);
}

template <typename T>
int run_bwd(const tg_attn_bwd_desc* d, hipStream_t st) {
  AttnBwdParams p{};
  p.heads = d->heads; p.hd = d->head_dim; p.n_r = d->n; p.n_s = d->n; p.n_z = d->n; p.n_rblk = (d->n + 127) / 128;
  p.extra = nullptr; p.e_ld = 0; p.ds_scale = d->scale;
  p.r_ld = p.s_ld = p.o_ld = d->ld; p.r_bs = p.s_bs = p.o_bs = d->bs;
  p.z_ld = d->t_ld; p.z_bs = d->t_bs;
  p.stats = d->stats;
  p.scale = d->scale; p.scale_log2 = d->scale * 1.4426950408889634f;
  // statistics, then dQ: queries in registers, keys streamed
  p.r1 = d->q; p.r2 = d->dout; p.s1 = d->k; p.s2 = d->v; p.z1 = d->kt; p.z2 = d->kt;
  p.out1 = d->dq; p.out2 = nullptr;
  launch_bwd<T, 2>(p, d->batch, st);
  TG_LAUNCH_CHECK();
  launch_bwd<T, 0>(p, d->batch, st);
  TG_LAUNCH_CHECK();
  // dK, dV: keys in registers, queries streamed
  p.r1 = d->k; p.r2 = d->v; p.s1 = d->q; p.s2 = d->dout; p.z1 = d->qt; p.z2 = d->doutt;
  p.out1 = d->dk; p.out2 = d->dv;
  launch_bwd<T, 1>(p, d->batch, st);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

// cross-attention: the keys / values are constants of the conditioning — only dQ, over a short key set, optionally with the guidance term on the probabilities
template <typename T>
int run_bwd_cross(const tg_attn_bwd_cross_desc* d, hipStream_t st) {
  AttnBwdParams p{};
  p.heads = d->heads; p.hd = d->head_dim; p.n_r = d->n_q; p.n_s = d->n_k; p.n_z = (d->n_k + 7) & ~7; p.n_rblk = (d->n_q + 127) / 128;
  p.r_ld = p.o_ld = d->q_ld; p.r_bs = p.o_bs = d->q_bs;
  p.s_ld = d->k_ld; p.s_bs = d->k_bs;
  p.z_ld = d->t_ld; p.z_bs = d->t_bs;
  p.stats = d->stats;
  p.extra = d->extra; p.e_ld = d->extra_ld;
  p.scale = d->scale; p.scale_log2 = d->scale * 1.4426950408889634f; p.ds_scale = d->ds_scale;
  p.r1 = d->q; p.r2 = d->dout; p.s1 = d->k; p.s2 = d->v; p.z1 = d->kt; p.z2 = d->kt;
  p.out1 = d->dq; p.out2 = nullptr;
  launch_bwd<T, 2>(p, d->batch, st);
  TG_LAUNCH_CHECK();
  launch_bwd<T, 0>(p, d->batch, st);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

}  // namespace

extern "C" int tg_attention_bwd_cross(const tg_attn_bwd_cross_desc* d, void* stream) {
  TG_CHECK(d != nullptr, TG_ERR_ARG, "tg_attention_bwd_cross: null descriptor");
  TG_CHECK(d->dtype == TG_BF16 || d->dtype == TG_F16, TG_ERR_ARG, "tg_attention_bwd_cross: bad dtype");
  TG_CHECK(d->batch > 0 && d->heads > 0 && d->n_q > 0 && d->n_k > 0, TG_ERR_ARG, "tg_attention_bwd_cross: empty problem");
  TG_CHECK(d->head_dim > 0 && d->head_dim % 8 == 0 && d->head_dim <= 64, TG_ERR_UNSUPPORTED,
           "tg_attention_bwd_cross: head_dim %d unsupported (multiple of 8, <= 64)", d->head_dim);
  TG_CHECK(d->q && d->dout && d->k && d->v && d->kt && d->stats && d->dq, TG_ERR_ARG, "tg_attention_bwd_cross: null pointer");
  TG_CHECK(d->q_ld % 8 == 0 && d->k_ld % 8 == 0 && d->t_ld % 8 == 0 && d->t_ld >= ((d->n_k + 7) & ~7), TG_ERR_ARG,
           "tg_attention_bwd_cross: pitches must keep 16-byte alignment; the transposed keys are zero-padded to a multiple of 8 columns");
  TG_CHECK(d->extra == nullptr || d->extra_ld >= d->n_k, TG_ERR_ARG, "tg_attention_bwd_cross: extra rows shorter than the key set");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == TG_BF16) return run_bwd_cross<bf16_t>(d, st);
  return run_bwd_cross<f16_t>(d, st);
}

extern "C" int tg_attention_bwd(const tg_attn_bwd_desc* d, void* stream) {
  TG_CHECK(d != nullptr, TG_ERR_ARG, "tg_attention_bwd: null descriptor");
  TG_CHECK(d->dtype == TG_BF16 || d->dtype == TG_F16, TG_ERR_ARG, "tg_attention_bwd: bad dtype");
  TG_CHECK(d->batch > 0 && d->heads > 0 && d->n > 0, TG_ERR_ARG, "tg_attention_bwd: empty problem");
  TG_CHECK(d->head_dim > 0 && d->head_dim % 8 == 0 && d->head_dim <= 64, TG_ERR_UNSUPPORTED,
           "tg_attention_bwd: head_dim %d unsupported (multiple of 8, <= 64)", d->head_dim);
  TG_CHECK(d->n % 8 == 0, TG_ERR_UNSUPPORTED, "tg_attention_bwd: n (%d) must be a multiple of 8", d->n);
  TG_CHECK(d->q && d->k && d->v && d->dout && d->qt && d->kt && d->doutt && d->stats && d->dq && d->dk && d->dv, TG_ERR_ARG, "tg_attention_bwd: null pointer");
  TG_CHECK(d->ld % 8 == 0 && d->t_ld % 8 == 0 && d->ld >= (int64_t)d->heads * d->head_dim && d->t_ld >= d->n, TG_ERR_ARG,
           "tg_attention_bwd: pitches must keep 16-byte alignment and cover the rows");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == TG_BF16) return run_bwd<bf16_t>(d, st);
  return run_bwd<f16_t>(d, st);
}

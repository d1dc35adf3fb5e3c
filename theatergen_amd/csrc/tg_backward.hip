// Input-gradient kernels for `latent_backward_guidance` (reference models/pipelines.py:62-128: `torch.autograd.grad(loss,
// [latents])` through the UNet).  Only d loss / d INPUT is ever needed (weights are frozen), so the
// contractions of the backward pass are the forward MFMA kernels run on transposed / tap-flipped weights (tg_gemm) and this
// file holds what is left: the normalisation, activation and softmax Jacobians and the 2 x 2 sum behind a nearest upsample.
// All reductions are fixed-order (no atomics): the gradient is deterministic.  fp32 arithmetic, bf16 / fp16 storage.
// This path runs once per guidance iteration on ONE image (batch 1): kernels are written for clarity, not for the roofline.
#include "tg_common.h"

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  // fixed-order fold: lanes by xor-shuffle, the 4 waves by lane 0 in wave order
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ float silu_grad_f(float z) {
  const float s = 1.0f / (1.0f + __expf(-z));
  return s * (1.0f + z * (1.0f - s));
}

// GroupNorm (+ SiLU) backward wrt its input.  y = act(gamma * xhat + beta), xhat = (x - mean) * rstd over one (batch, group).
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = gamma * dy * act'(z).  One block per (batch item, group).
template <typename T>
__global__ __launch_bounds__(256) void groupnorm_bwd_kernel(const T* x, const T* dy, int hw, int C, int groups, float eps, const T* gamma,
                                                           const T* beta, int silu, T* dx) {
  __shared__ float red[4];
  const int b = blockIdx.x / groups, g = blockIdx.x % groups;
  const int cg = C / groups;
  const long n = (long)hw * cg;
  const T* xb = x + (long)b * hw * C + g * cg;
  const T* dyb = dy + (long)b * hw * C + g * cg;
  T* dxb = dx + (long)b * hw * C + g * cg;
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) s += to_f32<T>(xb[(i / cg) * C + (i % cg)]);
  const float mean = block_sum_256(s, red) / (float)n;
  float q = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) { const float d = to_f32<T>(xb[(i / cg) * C + (i % cg)]) - mean; q += d * d; }
  const float rstd = rsqrtf(block_sum_256(q, red) / (float)n + eps);
  float s1 = 0.f, s2 = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) {
    const int c = (int)(i % cg);
    const long o = (i / cg) * C + c;
    const float xh = (to_f32<T>(xb[o]) - mean) * rstd;
    const float ga = to_f32<T>(gamma[g * cg + c]);
    float gy = to_f32<T>(dyb[o]);
    if (silu) gy *= silu_grad_f(ga * xh + to_f32<T>(beta[g * cg + c]));
    gy *= ga;
    s1 += gy;
    s2 += gy * xh;
  }
  const float m1 = block_sum_256(s1, red) / (float)n;
  const float m2 = block_sum_256(s2, red) / (float)n;
  for (long i = threadIdx.x; i < n; i += 256) {
    const int c = (int)(i % cg);
    const long o = (i / cg) * C + c;
    const float xh = (to_f32<T>(xb[o]) - mean) * rstd;
    const float ga = to_f32<T>(gamma[g * cg + c]);
    float gy = to_f32<T>(dyb[o]);
    if (silu) gy *= silu_grad_f(ga * xh + to_f32<T>(beta[g * cg + c]));
    gy *= ga;
    dxb[o] = from_f32<T>(rstd * (gy - m1 - xh * m2));
  }
}

// ---- round 3: the same Jacobian spread over the chip --------------------------------------------------------------------
// groupnorm_bwd_kernel above gives ONE workgroup to each (batch item, group): 32 workgroups for the batch-1 backward pass of
// `latent_backward_guidance`, four scalar passes over strided 2-byte elements (222 us per launch on the 768^2 plan, a quarter
// of the whole iteration).  Here the rows of a batch item are cut into slabs and a workgroup owns (batch item, slab) over ALL
// channels with 16-byte accesses, in three launches that share one loop skeleton:
//   PHASE 0  sum x, sum x^2 per (slab, group)                                   -> part[b][slab][g][0..1]
//   PHASE 1  fold the statistics, then sum gy, sum gy * xhat per (slab, group)  -> part[b][slab][g][2..3]
//   PHASE 2  fold both, dx = rstd * (gy - mean(gy) - xhat * mean(gy * xhat))
// gy = gamma * dy * act'(gamma * xhat + beta).  Thread (tr, tc) walks rows tr, tr + TR, ... of the slab and the 8-channel chunks
// tc, tc + 256 of every row; per-channel partial sums go through LDS and are folded per group in a FIXED order (channel-major,
// then thread row), slabs are folded in fp64 in slab order: deterministic, no atomics.
constexpr int GNB_MAX_SLOTS = 2;              // 8-channel chunks per thread and row: C <= 4096

__host__ __device__ inline int gnb_slabs(int batch, long hw) {
  long s = (hw + 31) / 32;                    // at least 32 rows per slab
  const long cap = batch >= 128 ? 1 : 128 / batch;   // every workgroup of phases 1 / 2 folds the slabs of its item again (one thread per group,
  if (s > cap) s = cap;                       // fp64, slab order): 256 slabs made that fold longer than the row loop (33 - 41 us per launch)
  return s < 1 ? 1 : (int)s;
}

template <typename T, int PHASE>
__global__ __launch_bounds__(256) void gn_bwd_slab_kernel(const T* x, const T* dy, long hw, int C, int groups, float eps, const T* gamma,
                                                         const T* beta, int silu, T* dx, float* part, int nslab) {
  typedef typename Vec<T>::v8 V8;
  extern __shared__ float gnb_smem[];
  const int nch = C >> 3, cg = C / groups;
  const int TC = nch < 256 ? nch : 256, TR = 256 / TC;
  const int tid = threadIdx.x, tc = tid % TC, tr = tid / TC;
  const bool active = tr < TR;
  const int b = blockIdx.x / nslab, slab = blockIdx.x % nslab;
  const long rows_per = (hw + nslab - 1) / nslab;
  const long r0 = (long)slab * rows_per, r1 = r0 + rows_per < hw ? r0 + rows_per : hw;
  float* sstat = gnb_smem;                    // [4][groups]: mean, rstd, m1, m2
  float* spart = gnb_smem + 4 * groups;       // [2][TR][C]
  const double inv_n = 1.0 / ((double)hw * cg);
  float* pb = part + (long)b * nslab * groups * 4;
  if (PHASE >= 1) {
    if (tid < groups) {
      double sx = 0.0, sq = 0.0, s1 = 0.0, s2 = 0.0;
      for (int z = 0; z < nslab; ++z) {
        const float* q = pb + ((long)z * groups + tid) * 4;
        sx += q[0]; sq += q[1];
        if (PHASE == 2) { s1 += q[2]; s2 += q[3]; }
      }
      const double mean = sx * inv_n;
      double var = sq * inv_n - mean * mean;
      if (var < 0.0) var = 0.0;
      sstat[tid] = (float)mean;
      sstat[groups + tid] = (float)(1.0 / sqrt(var + (double)eps));
      sstat[2 * groups + tid] = (float)(s1 * inv_n);
      sstat[3 * groups + tid] = (float)(s2 * inv_n);
    }
    __syncthreads();
  }
  float a0[GNB_MAX_SLOTS][8], a1[GNB_MAX_SLOTS][8];
#pragma unroll
  for (int sl = 0; sl < GNB_MAX_SLOTS; ++sl)
#pragma unroll
    for (int e = 0; e < 8; ++e) { a0[sl][e] = 0.f; a1[sl][e] = 0.f; }
#pragma unroll
  for (int sl = 0; sl < GNB_MAX_SLOTS; ++sl) {
    const int ch = tc + sl * 256;
    if (!active || ch >= nch) continue;
    const int c0 = ch * 8;
    float mean8[8], rstd8[8], ga8[8], be8[8], m18[8], m28[8];
    if (PHASE >= 1) {
      const V8 gv = *reinterpret_cast<const V8*>(gamma + c0);
      const V8 bv = *reinterpret_cast<const V8*>(beta + c0);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int g = (c0 + e) / cg;
        mean8[e] = sstat[g]; rstd8[e] = sstat[groups + g]; m18[e] = sstat[2 * groups + g]; m28[e] = sstat[3 * groups + g];
        ga8[e] = to_f32<T>(gv[e]); be8[e] = to_f32<T>(bv[e]);
      }
    }
    for (long r = r0 + tr; r < r1; r += TR) {
      const long o = ((long)b * hw + r) * C + c0;
      const V8 xv = *reinterpret_cast<const V8*>(x + o);
      if (PHASE == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float v = to_f32<T>(xv[e]); a0[sl][e] += v; a1[sl][e] += v * v; }
      } else {
        const V8 dv = *reinterpret_cast<const V8*>(dy + o);
        V8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (to_f32<T>(xv[e]) - mean8[e]) * rstd8[e];
          float gy = to_f32<T>(dv[e]);
          if (silu) gy *= silu_grad_f(ga8[e] * xh + be8[e]);
          gy *= ga8[e];
          if (PHASE == 1) { a0[sl][e] += gy; a1[sl][e] += gy * xh; }
          else ov[e] = from_f32<T>(rstd8[e] * (gy - m18[e] - xh * m28[e]));
        }
        if (PHASE == 2) *reinterpret_cast<V8*>(dx + o) = ov;
      }
    }
    if (PHASE < 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        spart[(long)tr * C + c0 + e] = a0[sl][e];
        spart[(long)(TR + tr) * C + c0 + e] = a1[sl][e];
      }
    }
  }
  if (PHASE < 2) {
    __syncthreads();
    if (tid < groups) {
      float s0 = 0.f, s1 = 0.f;
      for (int c = tid * cg; c < (tid + 1) * cg; ++c)
        for (int t = 0; t < TR; ++t) { s0 += spart[(long)t * C + c]; s1 += spart[(long)(TR + t) * C + c]; }
      float* q = pb + ((long)slab * groups + tid) * 4;
      q[PHASE == 0 ? 0 : 2] = s0;
      q[PHASE == 0 ? 1 : 3] = s1;
    }
  }
}

template <typename T>
void launch_gn_bwd_slabs(const T* x, const T* dy, int batch, long hw, int C, int groups, float eps, const T* gamma, const T* beta, int silu,
                         T* dx, float* part, hipStream_t st) {
  const int nslab = gnb_slabs(batch, hw);
  const int nch = C >> 3, TC = nch < 256 ? nch : 256, TR = 256 / TC;
  const size_t lds = ((size_t)4 * groups + (size_t)2 * TR * C) * sizeof(float);
  const dim3 grid((unsigned)(batch * nslab));
  hipLaunchKernelGGL((gn_bwd_slab_kernel<T, 0>), grid, dim3(256), lds, st, x, dy, hw, C, groups, eps, gamma, beta, silu, dx, part, nslab);
  hipLaunchKernelGGL((gn_bwd_slab_kernel<T, 1>), grid, dim3(256), lds, st, x, dy, hw, C, groups, eps, gamma, beta, silu, dx, part, nslab);
  hipLaunchKernelGGL((gn_bwd_slab_kernel<T, 2>), grid, dim3(256), lds, st, x, dy, hw, C, groups, eps, gamma, beta, silu, dx, part, nslab);
}

// LayerNorm backward wrt its input: one wave per row.
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* x, const T* dy, long rows, int C, float eps, const T* gamma, T* dx) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * C;
  const T* dr = dy + row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += to_f32<T>(xr[c]);
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = to_f32<T>(xr[c]) - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float xh = (to_f32<T>(xr[c]) - mean) * rstd;
    const float gy = to_f32<T>(dr[c]) * (gamma ? to_f32<T>(gamma[c]) : 1.f);
    s1 += gy;
    s2 += gy * xh;
  }
  const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
  for (int c = lane; c < C; c += 64) {
    const float xh = (to_f32<T>(xr[c]) - mean) * rstd;
    const float gy = to_f32<T>(dr[c]) * (gamma ? to_f32<T>(gamma[c]) : 1.f);
    dx[row * C + c] = from_f32<T>(rstd * (gy - m1 - xh * m2));
  }
}

// GEGLU backward (models/attention.py:337-338: out = a * gelu(gate), h = [a | gate]): dh = [dg * gelu(gate) | dg * a * gelu'(gate)]
template <typename T>
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const T* h, const T* dg, long rows, long inner, T* dh) {
  const long total = rows * inner;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / inner, c = i - r * inner;
    const float a = to_f32<T>(h[r * 2 * inner + c]), gt = to_f32<T>(h[r * 2 * inner + inner + c]), d = to_f32<T>(dg[i]);
    const float cdf = 0.5f * (1.0f + erff(gt * 0.70710678118654752440f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * gt * gt);
    dh[r * 2 * inner + c] = from_f32<T>(d * gt * cdf);
    dh[r * 2 * inner + inner + c] = from_f32<T>(d * a * (cdf + gt * pdf));
  }
}

// softmax backward per row: dS = scale * P * (dP + extra - sum_j P_j (dP_j + extra_j)).  P fp32 [rows, L] (tg_attn_probs),
// dP storage dtype [rows, ld] (a GEMM output), `extra` (optional) fp32 [rows, L] = d loss / d P added by the guidance loss at the
// saved cross-attention maps.  Outputs in the storage dtype with row pitch ld_out >= L (pad columns zeroed): dS and (optionally) P.
template <typename T, typename PT>
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const PT* P, long ldp, const T* dP, long ld, const float* extra, long lde,
                                                              long rows, int L, float scale, T* dS, T* Pout, long ld_out) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const PT* pr = P + row * ldp;
  const T* dr = dP + row * ld;
  const float* er = extra ? extra + row * lde : nullptr;
  float s = 0.f;
  for (int j = lane; j < L; j += 64) s += (float)pr[j] * (to_f32<T>(dr[j]) + (er ? er[j] : 0.f));
  s = wave_sum(s);
  for (int j = lane; j < (int)ld_out; j += 64) {
    float v = 0.f, pv = 0.f;
    if (j < L) {
      pv = (float)pr[j];
      v = scale * pv * (to_f32<T>(dr[j]) + (er ? er[j] : 0.f) - s);
    }
    dS[row * ld_out + j] = from_f32<T>(v);
    if (Pout) Pout[row * ld_out + j] = from_f32<T>(pv);
  }
}

// backward of the nearest x2 upsample in front of Upsample2D's conv: out[b, y, x, c] = sum of the 2 x 2 block of du
template <typename T>
__global__ __launch_bounds__(256) void sumpool2x2_kernel(const T* du, int batch, int h, int w, int C, T* out) {
  const long total = (long)batch * h * w * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long pix = i / C;
    const int x = (int)(pix % w), y = (int)((pix / w) % h);
    const long b = pix / ((long)w * h);
    const T* base = du + ((b * 2 * h + 2 * y) * (2L * w) + 2 * x) * C + c;
    const float v = (to_f32<T>(base[0]) + to_f32<T>(base[C])) + (to_f32<T>(base[2L * w * C]) + to_f32<T>(base[2L * w * C + C]));
    out[i] = from_f32<T>(v);
  }
}

inline unsigned grid_of(long n) {
  long g = (n + 255) / 256;
  return (unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int64_t tg_groupnorm_bwd_scratch_bytes(int32_t batch, int64_t hw, int32_t groups) {
  if (batch <= 0 || hw <= 0 || groups <= 0) return 0;
  return (int64_t)batch * gnb_slabs(batch, hw) * groups * 4 * (int64_t)sizeof(float);
}

extern "C" int tg_groupnorm_bwd(int32_t dtype, const void* x, const void* dy, int32_t batch, int64_t hw, int32_t channels, int32_t groups, float eps,
                                const void* gamma, const void* beta, int32_t silu, void* dx, void* partials, void* stream) {
  TG_CHECK((dtype == TG_BF16 || dtype == TG_F16) && x && dy && dx && gamma && beta, TG_ERR_ARG, "tg_groupnorm_bwd: bad args");
  TG_CHECK(batch > 0 && hw > 0 && groups > 0 && channels % groups == 0, TG_ERR_ARG, "tg_groupnorm_bwd: bad shape");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  // slab kernels: whole 16-byte channel chunks, at most two per thread and row, one folding thread per group; anything else (and a
  // call without scratch) keeps the one-workgroup-per-group kernel
  const bool slabs = partials != nullptr && channels % 8 == 0 && channels <= 8 * 256 * GNB_MAX_SLOTS && groups <= 256 && al16(x) && al16(dy) &&
                     al16(dx) && al16(gamma) && al16(beta);
  if (slabs) {
    if (dtype == TG_BF16)
      launch_gn_bwd_slabs<bf16_t>((const bf16_t*)x, (const bf16_t*)dy, batch, (long)hw, channels, groups, eps, (const bf16_t*)gamma, (const bf16_t*)beta, silu,
                                  (bf16_t*)dx, (float*)partials, st);
    else
      launch_gn_bwd_slabs<f16_t>((const f16_t*)x, (const f16_t*)dy, batch, (long)hw, channels, groups, eps, (const f16_t*)gamma, (const f16_t*)beta, silu,
                                 (f16_t*)dx, (float*)partials, st);
    TG_LAUNCH_CHECK();
    return TG_OK;
  }
  if (dtype == TG_BF16)
    hipLaunchKernelGGL(groupnorm_bwd_kernel<bf16_t>, dim3((unsigned)(batch * groups)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, (int)hw,
                       channels, groups, eps, (const bf16_t*)gamma, (const bf16_t*)beta, silu, (bf16_t*)dx);
  else
    hipLaunchKernelGGL(groupnorm_bwd_kernel<f16_t>, dim3((unsigned)(batch * groups)), dim3(256), 0, st, (const f16_t*)x, (const f16_t*)dy, (int)hw,
                       channels, groups, eps, (const f16_t*)gamma, (const f16_t*)beta, silu, (f16_t*)dx);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_layernorm_bwd(int32_t dtype, const void* x, const void* dy, int64_t rows, int32_t channels, float eps, const void* gamma,
                                void* dx, void* stream) {
  TG_CHECK((dtype == TG_BF16 || dtype == TG_F16) && x && dy && dx && rows > 0 && channels > 0, TG_ERR_ARG, "tg_layernorm_bwd: bad args");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == TG_BF16)
    hipLaunchKernelGGL(layernorm_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, (long)rows, channels, eps,
                       (const bf16_t*)gamma, (bf16_t*)dx);
  else
    hipLaunchKernelGGL(layernorm_bwd_kernel<f16_t>, grid, dim3(256), 0, st, (const f16_t*)x, (const f16_t*)dy, (long)rows, channels, eps,
                       (const f16_t*)gamma, (f16_t*)dx);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_geglu_bwd(int32_t dtype, const void* h, const void* dg, int64_t rows, int64_t inner, void* dh, void* stream) {
  TG_CHECK((dtype == TG_BF16 || dtype == TG_F16) && h && dg && dh && rows > 0 && inner > 0, TG_ERR_ARG, "tg_geglu_bwd: bad args");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == TG_BF16)
    hipLaunchKernelGGL(geglu_bwd_kernel<bf16_t>, dim3(grid_of(rows * inner)), dim3(256), 0, st, (const bf16_t*)h, (const bf16_t*)dg, (long)rows, (long)inner, (bf16_t*)dh);
  else
    hipLaunchKernelGGL(geglu_bwd_kernel<f16_t>, dim3(grid_of(rows * inner)), dim3(256), 0, st, (const f16_t*)h, (const f16_t*)dg, (long)rows, (long)inner, (f16_t*)dh);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_softmax_bwd_rows(int32_t dtype, const void* probs, int32_t probs_fp32, int64_t ld_probs, const void* dprobs, int64_t ld_dprobs,
                                   const float* extra, int64_t ld_extra, int64_t rows, int32_t length, float scale, void* dscores, void* probs_out,
                                   int64_t ld_out, void* stream) {
  TG_CHECK((dtype == TG_BF16 || dtype == TG_F16) && probs && dprobs && dscores && rows > 0 && length > 0, TG_ERR_ARG, "tg_softmax_bwd_rows: bad args");
  TG_CHECK(ld_probs >= length && ld_dprobs >= length && ld_out >= length && (!extra || ld_extra >= length), TG_ERR_ARG, "tg_softmax_bwd_rows: bad pitches");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((rows + 3) / 4));
#define TG_SMB(T, PT)                                                                                                                          \
  hipLaunchKernelGGL((softmax_bwd_rows_kernel<T, PT>), grid, dim3(256), 0, st, (const PT*)probs, (long)ld_probs, (const T*)dprobs, (long)ld_dprobs, \
                     extra, (long)ld_extra, (long)rows, length, scale, (T*)dscores, (T*)probs_out, (long)ld_out)
  if (dtype == TG_BF16) {
    if (probs_fp32) TG_SMB(bf16_t, float); else TG_SMB(bf16_t, bf16_t);
  } else {
    if (probs_fp32) TG_SMB(f16_t, float); else TG_SMB(f16_t, f16_t);
  }
#undef TG_SMB
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_sumpool2x2(int32_t dtype, const void* du, int32_t batch, int32_t h, int32_t w, int32_t channels, void* out, void* stream) {
  TG_CHECK((dtype == TG_BF16 || dtype == TG_F16) && du && out && batch > 0 && h > 0 && w > 0 && channels > 0, TG_ERR_ARG, "tg_sumpool2x2: bad args");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long n = (long)batch * h * w * channels;
  if (dtype == TG_BF16)
    hipLaunchKernelGGL(sumpool2x2_kernel<bf16_t>, dim3(grid_of(n)), dim3(256), 0, st, (const bf16_t*)du, batch, h, w, channels, (bf16_t*)out);
  else
    hipLaunchKernelGGL(sumpool2x2_kernel<f16_t>, dim3(grid_of(n)), dim3(256), 0, st, (const f16_t*)du, batch, h, w, channels, (f16_t*)out);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

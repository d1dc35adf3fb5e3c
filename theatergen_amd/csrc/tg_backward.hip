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

extern "C" int tg_groupnorm_bwd(int32_t dtype, const void* x, const void* dy, int32_t batch, int64_t hw, int32_t channels, int32_t groups,
                                float eps, const void* gamma, const void* beta, int32_t silu, void* dx, void* stream) {
  TG_CHECK((dtype == TG_BF16 || dtype == TG_F16) && x && dy && dx && gamma && beta, TG_ERR_ARG, "tg_groupnorm_bwd: bad args");
  TG_CHECK(batch > 0 && hw > 0 && groups > 0 && channels % groups == 0, TG_ERR_ARG, "tg_groupnorm_bwd: bad shape");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
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

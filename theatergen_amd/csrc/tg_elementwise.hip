// Elementwise / boundary kernels (HBM- or latency-bound): GEGLU gate, activations, residual add,
// conv_in / conv_out (tiny channel counts, NCHW <-> token-major conversion folded in), sinusoidal
// timestep embedding, the fused CFG + DDIM + frozen-mask step epilogue, latent blend / shift / compose.
#include "tg_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void geglu_kernel(const T* x, long rows, long inner, T* out) {
  typedef typename Vec<T>::v8 V8;
  const long c8 = inner / 8;
  const long total = rows * c8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / c8, c = (i - r * c8) * 8;
    V8 a = *reinterpret_cast<const V8*>(x + r * 2 * inner + c);
    V8 g = *reinterpret_cast<const V8*>(x + r * 2 * inner + inner + c);
    V8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = from_f32<T>(to_f32<T>(a[j]) * gelu_erf_f(to_f32<T>(g[j])));
    *reinterpret_cast<V8*>(out + r * inner + c) = o;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void act_kernel(const T* x, long n, int act, T* out) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float f = to_f32<T>(x[i]);
    f = apply_act(f, act);
    out[i] = from_f32<T>(f);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* a, const T* b, long n, T* out) {
  typedef typename Vec<T>::v8 V8;
  const long n8 = n / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    V8 x = *reinterpret_cast<const V8*>(a + i * 8), y = *reinterpret_cast<const V8*>(b + i * 8), o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = from_f32<T>(to_f32<T>(x[j]) + to_f32<T>(y[j]));
    *reinterpret_cast<V8*>(out + i * 8) = o;
  }
  for (long i = n8 * 8 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = from_f32<T>(to_f32<T>(a[i]) + to_f32<T>(b[i]));
}

__device__ __forceinline__ float load_src(const void* p, int src_dtype, long i) {
  if (src_dtype == 0) return (float)reinterpret_cast<const bf16_t*>(p)[i];
  if (src_dtype == 1) return (float)reinterpret_cast<const f16_t*>(p)[i];
  return reinterpret_cast<const float*>(p)[i];
}

// one thread = one (pixel, output channel): cin is tiny (4), K = 9*cin
template <typename T>
__global__ __launch_bounds__(256) void conv_in_kernel(const void* sample, int src_dtype, int batch, int cin, int h, int w,
                                                      const T* weight, const T* bias, int cout, T* out) {
  extern __shared__ float patch[];  // [pixels_per_block][9*cin]
  const int K = 9 * cin;
  const int ppb = blockDim.x / 64;            // pixels per block (64 threads cooperate per pixel)
  const long pix0 = (long)blockIdx.x * ppb;
  const long npix = (long)batch * h * w;
  for (int i = threadIdx.x; i < ppb * K; i += blockDim.x) {
    const int lp = i / K, k = i - lp * K;
    const long pix = pix0 + lp;
    float v = 0.f;
    if (pix < npix) {
      const int tap = k / cin, c = k - tap * cin;
      const int ky = tap / 3, kx = tap - ky * 3;
      const int b = (int)(pix / (h * w));
      const int r = (int)(pix - (long)b * h * w);
      const int y = r / w + ky - 1, x = r % w + kx - 1;
      if (y >= 0 && y < h && x >= 0 && x < w) v = load_src(sample, src_dtype, (((long)b * cin + c) * h + y) * w + x);
    }
    patch[i] = v;
  }
  __syncthreads();
  const int lp = threadIdx.x / 64, l = threadIdx.x & 63;
  const long pix = pix0 + lp;
  if (pix >= npix) return;
  for (int n = l; n < cout; n += 64) {
    float acc = bias ? to_f32<T>(bias[n]) : 0.f;
    for (int k = 0; k < K; ++k) acc += patch[lp * K + k] * to_f32<T>(weight[(long)n * K + k]);
    out[pix * cout + n] = from_f32<T>(acc);
  }
}

// one wave = one output pixel: 64 lanes split K = 9*cin, all cout (<= 8) outputs reduced by shuffles
template <typename T>
__global__ __launch_bounds__(256) void conv_out_kernel(const T* x, int batch, int cin, int h, int w, const T* weight,
                                                       const T* bias, int cout, void* out, int out_f32) {
  typedef typename Vec<T>::v8 V8;
  const int lane = threadIdx.x & 63;
  const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long npix = (long)batch * h * w;
  if (pix >= npix) return;
  const int b = (int)(pix / (h * w));
  const int r = (int)(pix - (long)b * h * w);
  const int oy = r / w, ox = r % w;
  const int c8 = cin / 8;
  float acc[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) acc[n] = 0.f;
  for (int i = lane; i < 9 * c8; i += 64) {
    const int tap = i / c8, c = (i - tap * c8) * 8;
    const int y = oy + tap / 3 - 1, xx = ox + tap % 3 - 1;
    if (y < 0 || y >= h || xx < 0 || xx >= w) continue;
    V8 v = *reinterpret_cast<const V8*>(x + (((long)b * h + y) * w + xx) * cin + c);
    for (int n = 0; n < cout; ++n) {
      V8 wv = *reinterpret_cast<const V8*>(weight + ((long)n * 9 + tap) * cin + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[n] += to_f32<T>(v[j]) * to_f32<T>(wv[j]);
    }
  }
  for (int n = 0; n < cout; ++n) {
    float s = wave_sum(acc[n]);
    if (lane == 0) {
      s += bias ? to_f32<T>(bias[n]) : 0.f;
      const long o = (((long)b * cout + n) * h + oy) * w + ox;
      if (out_f32) reinterpret_cast<float*>(out)[o] = s;
      else reinterpret_cast<T*>(out)[o] = from_f32<T>(s);
    }
  }
}

template <typename T>
__global__ void timestep_embedding_kernel(const float* t, const int* index, int t_stride, int rows, int dim, int flip,
                                          float freq_shift, T* out, long ldo) {
  const int half = dim / 2;
  const int r = blockIdx.x;
  const float tv = t[(index ? *index : 0) + (long)r * t_stride];
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    const float e = __expf(-9.210340371976184f * (float)i / ((float)half - freq_shift));  // ln(10000)
    const float a = tv * e;
    const float s = sinf(a), c = cosf(a);
    T* o = out + (long)r * ldo;
    if (flip) { o[i] = from_f32<T>(c); o[half + i] = from_f32<T>(s); }
    else { o[i] = from_f32<T>(s); o[half + i] = from_f32<T>(c); }
  }
}

struct StepParams {
  const float* noise_pred;
  float* latents;
  int n_img, chw, hw;
  int has_cfg;
  float g;
  const float* coef;
  int* step_idx;
  int advance;
  int pred_type;
  const float* frozen;
  const float* frozen_mask;
  int mask_per_img;
  int frozen_steps;
  float* history;
  void* model_in;
  int model_in_dtype;
};

__global__ __launch_bounds__(256) void step_epilogue_kernel(StepParams p) {
  const int step = *p.step_idx;
  const float sa = p.coef[step * 4], sb = p.coef[step * 4 + 1], sap = p.coef[step * 4 + 2], sbp = p.coef[step * 4 + 3];
  const long total = (long)p.n_img * p.chw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float u = p.noise_pred[i];
    float mo = u;
    if (p.has_cfg) { const float c = p.noise_pred[total + i]; mo = u + p.g * (c - u); }
    const float x = p.latents[i];
    float x0, eps;
    if (p.pred_type == 0) { x0 = (x - sb * mo) / sa; eps = mo; }
    else { x0 = sa * x - sb * mo; eps = sa * mo + sb * x; }
    float nx = sap * x0 + sbp * eps;
    if (p.frozen && step < p.frozen_steps) {
      const long img = i / p.chw;
      const long pix = (i - img * p.chw) % p.hw;
      const float m = p.frozen_mask[(p.mask_per_img ? img * p.hw : 0) + pix];
      const float f = p.frozen[(long)(step + 1) * total + i];
      nx = f * m + nx * (1.f - m);
    }
    p.latents[i] = nx;
    if (p.history) p.history[(long)(step + 1) * total + i] = nx;
    if (p.model_in) {
      // next UNet input = cat([latents] * 2) in the model dtype (models/pipelines.py:409-414)
      if (p.model_in_dtype == TG_BF16) {
        reinterpret_cast<bf16_t*>(p.model_in)[i] = (bf16_t)nx;
        reinterpret_cast<bf16_t*>(p.model_in)[total + i] = (bf16_t)nx;
      } else if (p.model_in_dtype == TG_F16) {
        reinterpret_cast<f16_t*>(p.model_in)[i] = (f16_t)nx;
        reinterpret_cast<f16_t*>(p.model_in)[total + i] = (f16_t)nx;
      } else {
        reinterpret_cast<float*>(p.model_in)[i] = nx;
        reinterpret_cast<float*>(p.model_in)[total + i] = nx;
      }
    }
  }
}

__global__ void step_advance_kernel(int* step_idx) { *step_idx += 1; }

// ROUND: 0 = fp32 latents (the reference with an fp32 UNet); 1 / 2 = the reference's arithmetic when the latents are
// fp16 / bf16 tensors (utils/latents.py:156-166 with dtype = unet.dtype, generate.py:77-81): `bg * sqrt(1-r)`,
// `fg * sqrt(r)` and their sum are half-precision tensor ops (each computed in fp32 and rounded to the storage type),
// the products with the fp32 mask promote to fp32, `.to(dtype)` rounds once more.  The fp32 output then holds values
// that are exactly representable in the storage type.
template <int ROUND>
__device__ __forceinline__ float round_storage(float x) {
  if (ROUND == 1) return (float)(f16_t)x;
  if (ROUND == 2) return (float)(bf16_t)x;
  return x;
}
template <int ROUND>
__global__ __launch_bounds__(256) void blend_kernel(const float* bg, const float* fg, const float* mask, int planes, int hw,
                                                    float s1, float s2, float sigma, float* out) {
  const long total = (long)planes * hw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float m = mask[i % hw];
    const float b = bg[i];
    const float inner = round_storage<ROUND>(round_storage<ROUND>(b * s1) + round_storage<ROUND>(fg[i] * s2));
    out[i] = round_storage<ROUND>(round_storage<ROUND>(b * (1.f - m) + inner * m) * sigma);
  }
}

__global__ __launch_bounds__(256) void gaussian_sample_kernel(const float* moments, const float* noise, int batch, int ch, int hw,
                                                              float scale, float* out) {
  const long total = (long)batch * ch * hw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / ((long)ch * hw), r = i - b * (long)ch * hw;
    const float mean = moments[b * 2 * ch * hw + r];
    float v = mean;
    if (noise != nullptr) {
      const float logvar = fminf(fmaxf(moments[b * 2 * ch * hw + (long)ch * hw + r], -30.f), 20.f);
      v = mean + expf(0.5f * logvar) * noise[i];
    }
    out[i] = scale * v;
  }
}

__global__ __launch_bounds__(256) void add_noise_kernel(const float* x0, const float* noise, const float* ca, const float* cb,
                                                        int steps, long n, float* out) {
  const long total = (long)steps * n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long s_ = i / n, r = i - s_ * n;
    out[i] = ca[s_] * x0[r] + cb[s_] * noise[r];
  }
}

__global__ __launch_bounds__(256) void shift_kernel(const float* src, long planes, int h, int w, int dx, int dy, float* dst) {
  const long total = planes * h * w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pl = i / (h * w);
    const int r = (int)(i - pl * h * w);
    const int y = r / w, x = r % w;
    const int sy = y - dy, sx = x - dx;
    float v = 0.f;
    if (sy >= 0 && sy < h && sx >= 0 && sx < w) v = src[pl * h * w + (long)sy * w + sx];
    dst[i] = v;
  }
}

__global__ __launch_bounds__(256) void compose_kernel(float* dst, const float* src, const float* mask, long planes, int hw) {
  const long total = planes * hw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float m = mask[i % hw];
    dst[i] = dst[i] * (1.f - m) + src[i] * m;
  }
}

// dst[b][c][r] = src[b][r][c]  (NCHW <-> token-major), 32x32 LDS tiles, pad 1 -> conflict-free
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* src, int rows, int cols, T* dst) {
  __shared__ T tile[32][33];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const T* s = src + (long)b * rows * cols;
  T* d = dst + (long)b * rows * cols;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = s[(long)(r0 + i) * cols + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < cols && r0 + tx < rows) d[(long)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

// The same for large maps (rows % 128 == 0, cols % 64 == 0, 16-byte aligned): 128 x 64 tiles, 16-byte global reads along the source
// rows, 64-byte contiguous pieces of the destination rows per thread.  Round 3: the reverse pass transposes dS and P of the long
// self-attention rows (9216 x 9216 at 768^2: 170 MB each) per head; the strided torch copy it used ran at ~1 TB/s (355 us).
template <typename T>
__global__ __launch_bounds__(256) void transpose_big_kernel(const T* src, int rows, int cols, T* dst) {
  typedef typename Vec<T>::v8 V8;
  constexpr int TR = 128, TCc = 64, PITCH = TCc + 2;     // 33 dwords per tile row: the column gather below is conflict-free
  __shared__ T tile[TR * PITCH];
  const int b = blockIdx.z;
  const long r0 = (long)blockIdx.y * TR, c0 = (long)blockIdx.x * TCc;
  const T* s = src + (long)b * rows * cols;
  T* d = dst + (long)b * rows * cols;
  const int tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int v = it * 256 + tid;                         // vector index: row v / 8, 8-column group v % 8
    const int r = v >> 3, cgp = v & 7;
    const V8 x = *reinterpret_cast<const V8*>(s + (r0 + r) * cols + c0 + cgp * 8);
#pragma unroll
    for (int e = 0; e < 8; e += 2) {                      // 4-byte LDS stores (PITCH is even, cgp * 8 + e is even)
      typedef T __attribute__((ext_vector_type(2))) V2;
      V2 p2; p2[0] = x[e]; p2[1] = x[e + 1];
      *reinterpret_cast<V2*>(&tile[r * PITCH + cgp * 8 + e]) = p2;
    }
  }
  __syncthreads();
  const int c = tid & 63, rg = tid >> 6;                  // destination row c0 + c, source rows rg * 32 .. + 32
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = tile[(rg * 32 + q * 8 + e) * PITCH + c];
    *reinterpret_cast<V8*>(d + (c0 + c) * rows + r0 + rg * 32 + q * 8) = o;
  }
}

inline int grid_for(long n, int per_thread = 1) {
  long b = (n / per_thread + 255) / 256;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}

// ---- fast paths for the UNet's two thin convolutions (the generic kernels above stay as the fallback) -----------
// conv_in (4 -> 320 at SD-1.5): fp32 weights transposed to [K][cout] in LDS once per (persistent) block; a thread owns
// 8 consecutive output channels of TWO neighbouring pixels, so one pair of ds_read_b128 weight reads feeds 16 FMAs
// and the result leaves as one 16-byte store per pixel.  The generic kernel re-read every weight from global memory
// for every pixel (137 us at 16 x 64 x 64; this one is write-bandwidth bound).
template <typename T>
__global__ __launch_bounds__(256) void conv_in_fast_kernel(const void* sample, int src_dtype, int batch, int cin, int h, int w,
                                                           const T* weight, const T* bias, int cout, T* out, int groups) {
  typedef typename Vec<T>::v8 V8;
  extern __shared__ float cin_smem[];
  const int K = 9 * cin;
  float* sw = cin_smem;                  // [K][cout]
  float* patch = sw + K * cout;          // [groups * 2][K]
  const int tid = threadIdx.x, G = cout >> 3;
  for (int i = tid; i < K * cout; i += blockDim.x) {
    const int n = i / K, k = i - n * K;
    sw[k * cout + n] = to_f32<T>(weight[i]);
  }
  const long npix = (long)batch * h * w;
  const int ppb = groups * 2;
  const int g = tid / G, cg = tid - g * G;
  float bias8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = (bias != nullptr && g < groups) ? to_f32<T>(bias[cg * 8 + e]) : 0.f;
  for (long base = (long)blockIdx.x * ppb; base < npix; base += (long)gridDim.x * ppb) {
    __syncthreads();
    for (int i = tid; i < ppb * K; i += blockDim.x) {
      const int lp = i / K, k = i - lp * K;
      const long pix = base + lp;
      float v = 0.f;
      if (pix < npix) {
        const int tap = k / cin, c = k - tap * cin;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int b = (int)(pix / (h * w));
        const int r = (int)(pix - (long)b * h * w);
        const int y = r / w + ky - 1, x = r % w + kx - 1;
        if (y >= 0 && y < h && x >= 0 && x < w) v = load_src(sample, src_dtype, (((long)b * cin + c) * h + y) * w + x);
      }
      patch[i] = v;
    }
    __syncthreads();
    if (g >= groups) continue;
    float a0[8], a1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a0[e] = bias8[e]; a1[e] = bias8[e]; }
    const float* p0 = patch + (2 * g) * K;
    const float* p1 = p0 + K;
    const float* wp = sw + cg * 8;
    for (int k = 0; k < K; ++k) {
      const f32x4 wlo = *reinterpret_cast<const f32x4*>(wp + k * cout);
      const f32x4 whi = *reinterpret_cast<const f32x4*>(wp + k * cout + 4);
      const float x0 = p0[k], x1 = p1[k];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a0[e] = __builtin_fmaf(x0, wlo[e], a0[e]);
        a0[4 + e] = __builtin_fmaf(x0, whi[e], a0[4 + e]);
        a1[e] = __builtin_fmaf(x1, wlo[e], a1[e]);
        a1[4 + e] = __builtin_fmaf(x1, whi[e], a1[4 + e]);
      }
    }
    const long pix0 = base + 2 * g;
    V8 o;
    if (pix0 < npix) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(a0[e]);
      *reinterpret_cast<V8*>(out + pix0 * cout + cg * 8) = o;
    }
    if (pix0 + 1 < npix) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(a1[e]);
      *reinterpret_cast<V8*>(out + (pix0 + 1) * cout + cg * 8) = o;
    }
  }
}

// conv_in on the matrix cores (round 5; cin = 4: K = 36 -> three k-steps of 16, cout a multiple of 160).  The fp32-FMA kernel above is LDS-read bound at ~70 us
// for a tensor that takes ~8 us to write (16 x 64 x 64 x 320).  A wave owns 32 consecutive pixels: B fragments = the pixels' 3 x 3 x 4 patches gathered from
// the NCHW sample — split into a storage-dtype head and tail (x = hi + lo, two MFMAs per fragment) so that an fp32 sample keeps ~16 mantissa bits, as the
// fp32 kernel kept all of them —; A fragments = the weight rows [cout][36] zero-padded to 48 in LDS (112-byte rows: conflict-free ds_read_b128); 160
// channels at a time (5 accumulator tiles); bias, rounding, then a per-wave LDS bounce so that every store instruction covers 320-byte runs of pixel rows.
template <typename T>
__global__ __launch_bounds__(256) void conv_in_mfma_kernel(const void* sample, int src_dtype, int batch, int h, int w, const T* weight, const T* bias,
                                                           int cout, T* out) {
  typedef typename Vec<T>::v8 V8;
  typedef typename Vec<T>::v4 V4;
  constexpr int CIN = 4, K = 36, KP = 48, WP = 56, OP = 168;        // LDS pitches (elements): weight rows 112 B, bounce rows 336 B
  extern __shared__ __attribute__((aligned(16))) char cim_smem[];
  T* sw = reinterpret_cast<T*>(cim_smem);                           // [cout][WP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  T* so = sw + (size_t)cout * WP + (size_t)wave * 32 * OP;          // [32][OP] per wave
  // weight rows (72 bytes) in 8-byte pieces, 12 per padded row (the last three are the zero padding of k = 36 .. 47)
  for (int i = tid; i < cout * (KP / 4); i += 256) {
    const int n = i / (KP / 4), c = i - n * (KP / 4);
    V4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = from_f32<T>(0.f);
    if (c < K / 4) v = *reinterpret_cast<const V4*>(weight + n * K + 4 * c);
    *reinterpret_cast<V4*>(sw + n * WP + 4 * c) = v;
  }
  __syncthreads();
  const long npix = (long)batch * h * w;
  for (long base = ((long)blockIdx.x * 4 + wave) * 32; base < npix; base += (long)gridDim.x * 128) {
    const long pix = base + l31;
    const bool ok = pix < npix;
    const long pp = ok ? pix : 0;
    const int b = (int)(pp / ((long)h * w));
    const int r = (int)(pp - (long)b * h * w);
    const int y = r / w, x = r - y * w;
    V8 bh[3], bl[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 16 * ks + 8 * hi + e, tap = k >> 2, c = k & 3;
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int yy = y + ky - 1, xx = x + kx - 1;
        float v = 0.f;
        if (ok && tap < 9 && yy >= 0 && yy < h && xx >= 0 && xx < w) v = load_src(sample, src_dtype, (((long)b * CIN + c) * h + yy) * w + xx);
        const T vh = from_f32<T>(v);
        bh[ks][e] = vh;
        bl[ks][e] = from_f32<T>(v - to_f32<T>(vh));
      }
    for (int half = 0; half < cout / 160; ++half) {
      f32x16 acc[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const V8 a = *reinterpret_cast<const V8*>(sw + (size_t)(half * 160 + 32 * j + l31) * WP + 16 * ks + 8 * hi);
          acc[j] = mfma32(a, bl[ks], acc[j]);       // tails first: small terms into the accumulator before the large ones
          acc[j] = mfma32(a, bh[ks], acc[j]);
        }
      }
      // accumulator layout: lane & 31 = pixel, register quad g of lane half hi = channels 32 j + 8 g + 4 hi .. + 3
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch = 32 * j + 8 * g + 4 * hi;
          V4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(acc[j][4 * g + e] + (bias != nullptr ? to_f32<T>(bias[half * 160 + ch + e]) : 0.f));
          *reinterpret_cast<V4*>(so + l31 * OP + ch) = o;
        }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 10; ++it) {
        const int q = it * 64 + lane, px = q / 20, cn = q - px * 20;
        const V8 v = *reinterpret_cast<const V8*>(so + px * OP + cn * 8);
        if (base + px < npix) *reinterpret_cast<V8*>(out + (base + px) * cout + half * 160 + cn * 8) = v;
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

__device__ __forceinline__ float dot8_acc(bf16x8 a, bf16x8 b, float acc) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bf16x2 x = {a[2 * i], a[2 * i + 1]}, y = {b[2 * i], b[2 * i + 1]};
    acc = __builtin_amdgcn_fdot2_f32_bf16(x, y, acc, false);
  }
  return acc;
}
__device__ __forceinline__ float dot8_acc(f16x8 a, f16x8 b, float acc) {
  typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f16x2 x = {a[2 * i], a[2 * i + 1]}, y = {b[2 * i], b[2 * i + 1]};
    acc = __builtin_amdgcn_fdot2(x, y, acc, false);
  }
  return acc;
}

// conv_out (320 -> 4): the whole [cout][9*cin] weight lives in LDS; a wave walks pixel PAIRS, its lanes split the 9*cin
// reduction in 16-byte chunks and multiply with v_dot2 (no bf16 -> fp32 conversions), one weight read feeds both
// pixels.  (The generic kernel: 274 us, every lane re-loading weights from global memory per chunk.)
template <typename T, int COUT>
__global__ __launch_bounds__(256) void conv_out_fast_kernel(const T* x, int batch, int cin, int h, int w, const T* weight,
                                                            const T* bias, int cout, void* out, int out_f32) {
  typedef typename Vec<T>::v8 V8;
  extern __shared__ __attribute__((aligned(16))) char cout_smem[];
  T* sw = reinterpret_cast<T*>(cout_smem);
  const int K = 9 * cin, c8 = cin >> 3;
  for (int i = threadIdx.x; i < cout * K / 8; i += blockDim.x)
    reinterpret_cast<V8*>(sw)[i] = reinterpret_cast<const V8*>(weight)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const long npix = (long)batch * h * w, npair = (npix + 1) >> 1;
  const long nw = (long)gridDim.x * 4;
  V8 zero8;
#pragma unroll
  for (int e = 0; e < 8; ++e) zero8[e] = from_f32<T>(0.f);
  for (long pp = (long)blockIdx.x * 4 + (threadIdx.x >> 6); pp < npair; pp += nw) {
    const long pix0 = 2 * pp, pix1 = pix0 + 1;
    const bool v1 = pix1 < npix;
    const int b0 = (int)(pix0 / (h * w)), r0 = (int)(pix0 - (long)b0 * h * w);
    const int y0 = r0 / w, x0 = r0 - y0 * w;
    const long q1 = v1 ? pix1 : pix0;
    const int b1 = (int)(q1 / (h * w)), r1 = (int)(q1 - (long)b1 * h * w);
    const int y1 = r1 / w, x1 = r1 - y1 * w;
    float acc0[COUT], acc1[COUT];
#pragma unroll
    for (int n = 0; n < COUT; ++n) { acc0[n] = 0.f; acc1[n] = 0.f; }
    for (int i = lane; i < 9 * c8; i += 64) {
      const int tap = i / c8, c = (i - tap * c8) << 3;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      const int ya = y0 + dy, xa = x0 + dx, yb = y1 + dy, xb = x1 + dx;
      V8 va = zero8, vb = zero8;
      if (ya >= 0 && ya < h && xa >= 0 && xa < w) va = *reinterpret_cast<const V8*>(x + (((long)b0 * h + ya) * w + xa) * cin + c);
      if (v1 && yb >= 0 && yb < h && xb >= 0 && xb < w) vb = *reinterpret_cast<const V8*>(x + (((long)b1 * h + yb) * w + xb) * cin + c);
#pragma unroll
      for (int n = 0; n < COUT; ++n) {
        if (n < cout) {
          const V8 wv = *reinterpret_cast<const V8*>(sw + n * K + tap * cin + c);
          acc0[n] = dot8_acc(va, wv, acc0[n]);
          acc1[n] = dot8_acc(vb, wv, acc1[n]);
        }
      }
    }
#pragma unroll
    for (int n = 0; n < COUT; ++n) {
      if (n >= cout) continue;
      const float s0 = wave_sum(acc0[n]), s1 = wave_sum(acc1[n]);
      if (lane == 0) {
        const float bn = bias ? to_f32<T>(bias[n]) : 0.f;
        const long o0 = (((long)b0 * cout + n) * h + y0) * w + x0;
        const long o1 = (((long)b1 * cout + n) * h + y1) * w + x1;
        if (out_f32) {
          reinterpret_cast<float*>(out)[o0] = s0 + bn;
          if (v1) reinterpret_cast<float*>(out)[o1] = s1 + bn;
        } else {
          reinterpret_cast<T*>(out)[o0] = from_f32<T>(s0 + bn);
          if (v1) reinterpret_cast<T*>(out)[o1] = from_f32<T>(s1 + bn);
        }
      }
    }
  }
}

// conv_out on the matrix cores (round 5; cin <= 320 a multiple of 64, cout <= 8, h % 8 == 0, w % 16 == 0), optionally with conv_norm_out + SiLU applied
// while the window is staged (`coef`: tg_groupnorm_coef's a / d per (image, channel); the same fp32 expression and rounding as tg_groupnorm, so the
// normalised tensor is never written or read: UNet2DConditionModel.forward's conv_norm_out -> conv_act -> conv_out, models/unet_2d_condition.py:1015-1018).
// The dot-product kernel above re-reads every pixel nine times from L2 and reduces 8 accumulators across the wave per pixel pair: ~80 us (+ 15 us of
// GroupNorm apply) for a 42 MB tensor.  Here a workgroup owns an 8 x 16-pixel tile: its 10 x 18-pixel window (all channels, 115 KB) goes through LDS once
// (128-byte swizzle on the 16-byte slot, key = (pixel >> 1) & 7: conflict-free ds_read_b128 for 16 consecutive pixels), the weights [cout][9 cin] sit next to
// it, and a wave's 32 pixels x 9 taps x cin / 16 k-steps run as 32 x 32 x 16 MFMAs whose A operand carries the cout <= 8 real rows (the other rows are
// zero registers — the matrix pipe has cycles to spare here, the LDS pipe does not).  Output: NCHW, 16 consecutive pixels per 64-byte run.
template <typename T>
__global__ __launch_bounds__(512) void conv_out_mfma_kernel(const T* x, const float* coef, int a_silu, int batch, int cin, int h, int w, const T* weight,
                                                            const T* bias, int cout, void* out, int out_f32) {
  typedef typename Vec<T>::v8 V8;
  constexpr int TH = 8, TW = 16, WW = TW + 2, WIN = (TH + 2) * WW;       // 180 window pixels
  constexpr int NT = 512;      // EIGHT waves stage the window (two per SIMD: one's GroupNorm + SiLU arithmetic — ~70 VALU cycles per element, the staging's
                               // bound — runs under the other's loads; four waves: 77 us per launch in the replayed step); waves 0 .. 3 then own the 128 pixels
  extern __shared__ __attribute__((aligned(16))) char com_smem[];
  const int K = 9 * cin, ch8 = cin >> 3;
  T* sx = reinterpret_cast<T*>(com_smem);                 // [WIN][cin], swizzled
  T* sw = sx + (size_t)WIN * cin;                         // [cout][K]
  float* sc = reinterpret_cast<float*>(sw + (size_t)cout * K);   // [2][cin] coefficients of this tile's image
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int tiles_x = w / TW, tiles_y = h / TH;
  const int tile = blockIdx.x;
  const int b = tile / (tiles_x * tiles_y), tr = tile - b * tiles_x * tiles_y;
  const int y0 = (tr / tiles_x) * TH, x0 = (tr % tiles_x) * TW;
  for (int i = tid; i < cout * K / 8; i += NT) reinterpret_cast<V8*>(sw)[i] = reinterpret_cast<const V8*>(weight)[i];
  if (coef != nullptr) {
    for (int i = tid; i < 2 * cin; i += NT) sc[i] = coef[(long)b * 2 * cin + i];
    __syncthreads();
  }
  // ---- window: global -> registers (-> GroupNorm + SiLU) -> LDS, NU 16-byte chunks in flight per thread (the staging is latency-bound: 28 chunks per
  // thread at cin = 320; four in flight measured 47 us for the whole launch against ~6 us of LDS-pipe time)
  constexpr int NU = 7;
  const T* xb = x + (long)b * h * w * cin;
  const int nchunk = WIN * ch8;
  for (int q0 = tid; q0 < nchunk; q0 += NU * NT) {
    V8 v[NU];
    int wp[NU], c[NU];
    bool ok[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int q = q0 + u * NT;
      wp[u] = q / ch8; c[u] = q - wp[u] * ch8;
      const int wy = wp[u] / WW, wx = wp[u] - wy * WW;
      const int yy = y0 + wy - 1, xx = x0 + wx - 1;
      ok[u] = q < nchunk && yy >= 0 && yy < h && xx >= 0 && xx < w;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[u][e] = from_f32<T>(0.f);
      if (ok[u]) v[u] = *reinterpret_cast<const V8*>(xb + ((long)yy * w + xx) * cin + c[u] * 8);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (q0 + u * NT >= nchunk) continue;
      if (coef != nullptr && ok[u]) {
        const float* ca = sc + c[u] * 8;
        const float* cd = sc + cin + c[u] * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = to_f32<T>(v[u][e]) * ca[e] + cd[e];       // tg_norm.hip gn_apply_kernel: the same expression, the same rounding
          v[u][e] = from_f32<T>(a_silu ? silu_f(f) : f);
        }
      }
      const int slot = (c[u] & ~7) | ((c[u] ^ (wp[u] >> 1)) & 7);
      *reinterpret_cast<V8*>(sx + (size_t)wp[u] * cin + slot * 8) = v[u];
    }
  }
  __syncthreads();
  if (wave >= 4) return;
  // ---- 32 pixels per wave (tile rows 2 wave, 2 wave + 1) x 9 taps x cin / 16 k-steps
  const int ty = 2 * wave + (l31 >> 4), tx = l31 & 15;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  V8 zero8;
#pragma unroll
  for (int e = 0; e < 8; ++e) zero8[e] = from_f32<T>(0.f);
  const bool arow = l31 < cout;
  const T* wrow = sw + (size_t)(arow ? l31 : 0) * K + 8 * hi;
  const int nks = cin >> 4;
  for (int tap = 0; tap < 9; ++tap) {
    const int ky = tap / 3, kx = tap - 3 * ky;
    const int wpx = (ty + ky) * WW + tx + kx, key = (wpx >> 1) & 7;
    const T* xrow = sx + (size_t)wpx * cin;
    const T* wt = wrow + tap * cin;
#pragma unroll 4
    for (int ks = 0; ks < nks; ++ks) {
      const int cc = 2 * ks + hi;
      const V8 bfrag = *reinterpret_cast<const V8*>(xrow + (((cc & ~7) | ((cc ^ key) & 7)) << 3));
      V8 afrag = zero8;
      if (arow) afrag = *reinterpret_cast<const V8*>(wt + 16 * ks);
      acc = mfma32(afrag, bfrag, acc);
    }
  }
  // accumulator: lane & 31 = pixel, register r of lane half hi = output channel 8 (r >> 2) + 4 hi + (r & 3): channels 0 .. 7 are registers 0 .. 3 of the two halves
  const int yy = y0 + ty, xx = x0 + tx;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = 4 * hi + r;
    if (n >= cout) continue;
    const float val = acc[r] + (bias != nullptr ? to_f32<T>(bias[n]) : 0.f);
    const long o = (((long)b * cout + n) * h + yy) * w + xx;
    if (out_f32) reinterpret_cast<float*>(out)[o] = val;
    else reinterpret_cast<T*>(out)[o] = from_f32<T>(val);
  }
}

// ---- VAE decode helpers -------------------------------------------------------
// 1x1 convolution on a thin NCHW tensor (post_quant_conv 4 -> 4 with the 1 / scaling_factor of
// `vae.decode(latents / vae.config.scaling_factor)`, models/pipelines.py:849-854, folded in): one thread per pixel.
__global__ __launch_bounds__(256) void conv1x1_nchw_kernel(const float* x, int batch, int cin, int cout, long hw, const float* w,
                                                           const float* bias, float in_scale, float* out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)batch * hw) return;
  const long b = idx / hw, p = idx - b * hw;
  float v[8];
  for (int c = 0; c < cin; ++c) v[c] = x[(b * cin + c) * hw + p] * in_scale;
  for (int o = 0; o < cout; ++o) {
    float a = bias ? bias[o] : 0.f;
    for (int c = 0; c < cin; ++c) a += w[o * cin + c] * v[c];
    out[(b * cout + o) * hw + p] = a;
  }
}

// row softmax in place semantics (out may alias x): one wave per row, fp32 math, scale applied to the logits.
// Used by the VAE's single-head d = 512 attention (scores = GEMM, softmax here, PV = GEMM; head dims > 160 are outside
// the fused flash kernel's register budget and the layer runs once per image, not per step).
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const T* x, long rows, int cols, long ldx, float scale, T* out, long ldo) {
  typedef typename Vec<T>::v8 V8;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * ldx;
  T* orow = out + row * ldo;
  const int c8 = cols / 8;
  float mx = -INFINITY;
  for (int i = lane; i < c8; i += 64) {
    V8 v = *reinterpret_cast<const V8*>(xr + i * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) mx = fmaxf(mx, to_f32<T>(v[j]));
  }
  mx = wave_max(mx) * scale;
  const float c = scale * 1.4426950408889634f, mc = mx * 1.4426950408889634f;
  float sum = 0.f;
  for (int i = lane; i < c8; i += 64) {
    V8 v = *reinterpret_cast<const V8*>(xr + i * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += __builtin_amdgcn_exp2f(__builtin_fmaf(to_f32<T>(v[j]), c, -mc));
  }
  const float inv = 1.f / wave_sum(sum);
  for (int i = lane; i < c8; i += 64) {
    V8 v = *reinterpret_cast<const V8*>(xr + i * 8);
    V8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = from_f32<T>(__builtin_amdgcn_exp2f(__builtin_fmaf(to_f32<T>(v[j]), c, -mc)) * inv);
    *reinterpret_cast<V8*>(orow + i * 8) = o;
  }
}

}  // namespace

#define DISPATCH(dtype, NAME, GRID, BLOCK, LDS, ...)                                            \
  do {                                                                                          \
    if ((dtype) == TG_BF16) hipLaunchKernelGGL(NAME<bf16_t>, GRID, BLOCK, LDS, st, __VA_ARGS__); \
    else hipLaunchKernelGGL(NAME<f16_t>, GRID, BLOCK, LDS, st, __VA_ARGS__);                     \
  } while (0)

extern "C" int tg_geglu(int32_t dtype, const void* x, int64_t rows, int64_t inner, void* out, void* stream) {
  TG_CHECK((dtype == TG_BF16 || dtype == TG_F16) && x && out && rows > 0 && inner > 0 && inner % 8 == 0, TG_ERR_ARG,
           "tg_geglu: bad args");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int grid = grid_for(rows * inner / 8);
  if (dtype == TG_BF16) hipLaunchKernelGGL(geglu_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, rows, inner, (bf16_t*)out);
  else hipLaunchKernelGGL(geglu_kernel<f16_t>, dim3(grid), dim3(256), 0, st, (const f16_t*)x, rows, inner, (f16_t*)out);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_act(int32_t dtype, const void* x, int64_t n, int32_t act, void* out, void* stream) {
  TG_CHECK((dtype == TG_BF16 || dtype == TG_F16) && x && out && n > 0, TG_ERR_ARG, "tg_act: bad args");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int grid = grid_for(n);
  if (dtype == TG_BF16) hipLaunchKernelGGL(act_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, n, act, (bf16_t*)out);
  else hipLaunchKernelGGL(act_kernel<f16_t>, dim3(grid), dim3(256), 0, st, (const f16_t*)x, n, act, (f16_t*)out);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_add(int32_t dtype, const void* a, const void* b, int64_t n, void* out, void* stream) {
  TG_CHECK((dtype == TG_BF16 || dtype == TG_F16) && a && b && out && n > 0, TG_ERR_ARG, "tg_add: bad args");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int grid = grid_for(n, 8);
  if (dtype == TG_BF16) hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, n, (bf16_t*)out);
  else hipLaunchKernelGGL(add_kernel<f16_t>, dim3(grid), dim3(256), 0, st, (const f16_t*)a, (const f16_t*)b, n, (f16_t*)out);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_conv_in(int32_t dtype, const void* sample, int32_t src_dtype, int32_t batch, int32_t cin, int32_t h,
                          int32_t w, const void* weight, const void* bias, int32_t cout, void* out, void* stream) {
  TG_CHECK((dtype == TG_BF16 || dtype == TG_F16) && sample && weight && out, TG_ERR_ARG, "tg_conv_in: bad args");
  TG_CHECK(src_dtype >= 0 && src_dtype <= 2 && batch > 0 && cin > 0 && cin <= 16 && cout > 0, TG_ERR_ARG, "tg_conv_in: bad shape");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long npix = (long)batch * h * w;
  {
    // matrix-core path (round 5): cin = 4, cout a multiple of 160 (dev A/B knob TG_CONV_IN_MFMA=0: the fp32-FMA kernels below)
    static const bool mfma_on = [] { const char* e = getenv("TG_CONV_IN_MFMA"); return !(e && e[0] == '0'); }();
    if (mfma_on && cin == 4 && cout % 160 == 0 && cout <= 640 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(weight) & 7) == 0) {
      const size_t lds = ((size_t)cout * 56 + 4 * 32 * 168) * 2;
      long nb = (npix + 127) / 128;
      if (nb > 1024) nb = 1024;
      if (dtype == TG_BF16) {
        auto k = conv_in_mfma_kernel<bf16_t>;
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)attr;
        hipLaunchKernelGGL(k, dim3((unsigned)nb), dim3(256), lds, st, sample, src_dtype, batch, h, w, (const bf16_t*)weight, (const bf16_t*)bias, cout, (bf16_t*)out);
      } else {
        auto k = conv_in_mfma_kernel<f16_t>;
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)attr;
        hipLaunchKernelGGL(k, dim3((unsigned)nb), dim3(256), lds, st, sample, src_dtype, batch, h, w, (const f16_t*)weight, (const f16_t*)bias, cout, (f16_t*)out);
      }
      TG_LAUNCH_CHECK();
      return TG_OK;
    }
  }
  {
    // fast path: 8 output channels x 2 pixels per thread, fp32 weights resident in LDS
    const int K = 9 * cin, G = cout / 8;
    const int groups = G > 0 ? 256 / G : 0;
    if (cout % 8 == 0 && groups >= 1 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
      const size_t lds = ((size_t)K * cout + (size_t)groups * 2 * K) * sizeof(float);
      if (lds <= 64 * 1024) {
        long nb = (npix + groups * 2 - 1) / (groups * 2);
        if (nb > 512) nb = 512;
        if (dtype == TG_BF16)
          hipLaunchKernelGGL(conv_in_fast_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), lds, st, sample, src_dtype, batch, cin, h, w, (const bf16_t*)weight, (const bf16_t*)bias, cout, (bf16_t*)out, groups);
        else
          hipLaunchKernelGGL(conv_in_fast_kernel<f16_t>, dim3((unsigned)nb), dim3(256), lds, st, sample, src_dtype, batch, cin, h, w, (const f16_t*)weight, (const f16_t*)bias, cout, (f16_t*)out, groups);
        TG_LAUNCH_CHECK();
        return TG_OK;
      }
    }
  }
  const int ppb = 4;
  const size_t lds = (size_t)ppb * 9 * cin * sizeof(float);
  dim3 grid((unsigned)((npix + ppb - 1) / ppb));
  if (dtype == TG_BF16)
    hipLaunchKernelGGL(conv_in_kernel<bf16_t>, grid, dim3(256), lds, st, sample, src_dtype, batch, cin, h, w, (const bf16_t*)weight, (const bf16_t*)bias, cout, (bf16_t*)out);
  else
    hipLaunchKernelGGL(conv_in_kernel<f16_t>, grid, dim3(256), lds, st, sample, src_dtype, batch, cin, h, w, (const f16_t*)weight, (const f16_t*)bias, cout, (f16_t*)out);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

namespace {
// the matrix-core kernel takes the problem (dev A/B knob TG_CONV_OUT_MFMA=0: the dot-product kernels)
bool conv_out_mfma_ok(int cin, int h, int w, int cout) {
  static const bool on = [] { const char* e = getenv("TG_CONV_OUT_MFMA"); return !(e && e[0] == '0'); }();
  return on && cin % 64 == 0 && cin <= 320 && cout > 0 && cout <= 8 && h % 8 == 0 && w % 16 == 0;
}
}  // namespace

extern "C" int tg_conv_out_takes_coef(int32_t cin, int32_t h, int32_t w, int32_t cout) { return conv_out_mfma_ok(cin, h, w, cout) ? 1 : 0; }

static int conv_out_impl(int32_t dtype, const void* x, const float* coef, int32_t a_silu, int32_t batch, int32_t cin, int32_t h, int32_t w,
                         const void* weight, const void* bias, int32_t cout, void* out, int32_t out_f32, void* stream) {
  TG_CHECK((dtype == TG_BF16 || dtype == TG_F16) && x && weight && out, TG_ERR_ARG, "tg_conv_out: bad args");
  TG_CHECK(batch > 0 && cin % 8 == 0 && cout > 0 && cout <= 8, TG_ERR_ARG, "tg_conv_out: bad shape cin=%d cout=%d", cin, cout);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long npix = (long)batch * h * w;
  if (conv_out_mfma_ok(cin, h, w, cout) && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(weight) & 15) == 0) {
    const size_t lds = ((size_t)180 * cin + (size_t)cout * 9 * cin) * 2 + (size_t)2 * cin * 4;
    const dim3 g((unsigned)(batch * (h / 8) * (w / 16)));
    if (dtype == TG_BF16) {
      auto k = conv_out_mfma_kernel<bf16_t>;
      static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)attr;
      hipLaunchKernelGGL(k, g, dim3(512), lds, st, (const bf16_t*)x, coef, a_silu, batch, cin, h, w, (const bf16_t*)weight, (const bf16_t*)bias, cout, out, out_f32);
    } else {
      auto k = conv_out_mfma_kernel<f16_t>;
      static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)attr;
      hipLaunchKernelGGL(k, g, dim3(512), lds, st, (const f16_t*)x, coef, a_silu, batch, cin, h, w, (const f16_t*)weight, (const f16_t*)bias, cout, out, out_f32);
    }
    TG_LAUNCH_CHECK();
    return TG_OK;
  }
  TG_CHECK(coef == nullptr, TG_ERR_UNSUPPORTED, "tg_conv_out_gn: the GroupNorm prologue needs a problem the matrix-core kernel takes (tg_conv_out_takes_coef)");
  {
    const size_t lds = (size_t)cout * 9 * cin * 2;
    if (lds <= 64 * 1024 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(weight) & 15) == 0) {
      long nb = ((npix + 1) / 2 + 3) / 4;
      if (nb > 1024) nb = 1024;
      const dim3 g((unsigned)nb);
      if (dtype == TG_BF16) {
        if (cout <= 4) hipLaunchKernelGGL((conv_out_fast_kernel<bf16_t, 4>), g, dim3(256), lds, st, (const bf16_t*)x, batch, cin, h, w, (const bf16_t*)weight, (const bf16_t*)bias, cout, out, out_f32);
        else hipLaunchKernelGGL((conv_out_fast_kernel<bf16_t, 8>), g, dim3(256), lds, st, (const bf16_t*)x, batch, cin, h, w, (const bf16_t*)weight, (const bf16_t*)bias, cout, out, out_f32);
      } else {
        if (cout <= 4) hipLaunchKernelGGL((conv_out_fast_kernel<f16_t, 4>), g, dim3(256), lds, st, (const f16_t*)x, batch, cin, h, w, (const f16_t*)weight, (const f16_t*)bias, cout, out, out_f32);
        else hipLaunchKernelGGL((conv_out_fast_kernel<f16_t, 8>), g, dim3(256), lds, st, (const f16_t*)x, batch, cin, h, w, (const f16_t*)weight, (const f16_t*)bias, cout, out, out_f32);
      }
      TG_LAUNCH_CHECK();
      return TG_OK;
    }
  }
  dim3 grid((unsigned)((npix + 3) / 4));
  if (dtype == TG_BF16)
    hipLaunchKernelGGL(conv_out_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, batch, cin, h, w, (const bf16_t*)weight, (const bf16_t*)bias, cout, out, out_f32);
  else
    hipLaunchKernelGGL(conv_out_kernel<f16_t>, grid, dim3(256), 0, st, (const f16_t*)x, batch, cin, h, w, (const f16_t*)weight, (const f16_t*)bias, cout, out, out_f32);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_conv_out(int32_t dtype, const void* x, int32_t batch, int32_t cin, int32_t h, int32_t w,
                           const void* weight, const void* bias, int32_t cout, void* out, int32_t out_f32, void* stream) {
  return conv_out_impl(dtype, x, nullptr, 0, batch, cin, h, w, weight, bias, cout, out, out_f32, stream);
}

extern "C" int tg_conv_out_gn(int32_t dtype, const void* x, const float* coef, int32_t a_silu, int32_t batch, int32_t cin, int32_t h, int32_t w,
                              const void* weight, const void* bias, int32_t cout, void* out, int32_t out_f32, void* stream) {
  TG_CHECK(coef != nullptr, TG_ERR_ARG, "tg_conv_out_gn: null coef");
  return conv_out_impl(dtype, x, coef, a_silu, batch, cin, h, w, weight, bias, cout, out, out_f32, stream);
}

extern "C" int tg_timestep_embedding(int32_t dtype, const float* t, const int32_t* index, int32_t t_stride, int32_t rows, int32_t dim,
                                     int32_t flip_sin_to_cos, float freq_shift, void* out, int64_t ldo, void* stream) {
  TG_CHECK((dtype == TG_BF16 || dtype == TG_F16) && t && out && rows > 0 && dim > 0 && dim % 2 == 0, TG_ERR_ARG,
           "tg_timestep_embedding: bad args");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == TG_BF16)
    hipLaunchKernelGGL(timestep_embedding_kernel<bf16_t>, dim3(rows), dim3(128), 0, st, t, index, t_stride, rows, dim, flip_sin_to_cos, freq_shift, (bf16_t*)out, ldo);
  else
    hipLaunchKernelGGL(timestep_embedding_kernel<f16_t>, dim3(rows), dim3(128), 0, st, t, index, t_stride, rows, dim, flip_sin_to_cos, freq_shift, (f16_t*)out, ldo);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_step_epilogue(const float* noise_pred, float* latents, int32_t n_img, int32_t chw, int32_t hw,
                                int32_t has_cfg, float guidance_scale, const float* coef, int32_t* step_idx, int32_t advance,
                                int32_t prediction_type, const float* frozen, const float* frozen_mask,
                                int32_t mask_per_img, int32_t frozen_steps, float* history, void* model_in,
                                int32_t model_in_dtype, void* stream) {
  TG_CHECK(noise_pred && latents && coef && step_idx && n_img > 0 && chw > 0 && hw > 0 && chw % hw == 0, TG_ERR_ARG,
           "tg_step_epilogue: bad args");
  TG_CHECK(prediction_type == 0 || prediction_type == 1, TG_ERR_ARG, "tg_step_epilogue: bad prediction type");
  TG_CHECK(!frozen || frozen_mask, TG_ERR_ARG, "tg_step_epilogue: frozen latents need a mask");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  StepParams p{noise_pred, latents, n_img, chw, hw, has_cfg, guidance_scale, coef, step_idx, advance, prediction_type, frozen,
               frozen_mask, mask_per_img, frozen_steps, history, model_in, model_in_dtype};
  hipLaunchKernelGGL(step_epilogue_kernel, dim3(grid_for((long)n_img * chw)), dim3(256), 0, st, p);
  TG_LAUNCH_CHECK();
  if (advance) {
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, st, step_idx);
    TG_LAUNCH_CHECK();
  }
  return TG_OK;
}

extern "C" int tg_blend_latents(const float* bg, const float* fg, const float* mask, int32_t planes, int32_t hw,
                                float ratio, float sigma, int32_t storage_dtype, float* out, void* stream) {
  TG_CHECK(bg && fg && mask && out && planes > 0 && hw > 0, TG_ERR_ARG, "tg_blend_latents: bad args");
  TG_CHECK(storage_dtype >= -1 && storage_dtype <= 1, TG_ERR_ARG, "tg_blend_latents: storage_dtype must be -1 (fp32), TG_BF16 or TG_F16");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const float s1 = (float)sqrt(1.0 - (double)ratio), s2 = (float)sqrt((double)ratio);
  const dim3 grid(grid_for((long)planes * hw));
  if (storage_dtype == TG_F16) hipLaunchKernelGGL(blend_kernel<1>, grid, dim3(256), 0, st, bg, fg, mask, planes, hw, s1, s2, sigma, out);
  else if (storage_dtype == TG_BF16) hipLaunchKernelGGL(blend_kernel<2>, grid, dim3(256), 0, st, bg, fg, mask, planes, hw, s1, s2, sigma, out);
  else hipLaunchKernelGGL(blend_kernel<0>, grid, dim3(256), 0, st, bg, fg, mask, planes, hw, s1, s2, sigma, out);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_gaussian_sample(const float* moments, const float* noise, int32_t batch, int32_t channels, int32_t hw,
                                  float scale, float* out, void* stream) {
  TG_CHECK(moments && out && batch > 0 && channels > 0 && hw > 0, TG_ERR_ARG, "tg_gaussian_sample: bad args");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gaussian_sample_kernel, dim3(grid_for((long)batch * channels * hw)), dim3(256), 0, st, moments, noise,
                     batch, channels, hw, scale, out);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_add_noise(const float* x0, const float* noise, const float* ca, const float* cb, int32_t steps, int64_t n,
                            float* out, void* stream) {
  TG_CHECK(x0 && noise && ca && cb && out && steps > 0 && n > 0, TG_ERR_ARG, "tg_add_noise: bad args");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(add_noise_kernel, dim3(grid_for((long)steps * n)), dim3(256), 0, st, x0, noise, ca, cb, steps, (long)n, out);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_shift(const float* src, int64_t planes, int32_t h, int32_t w, int32_t dx, int32_t dy, float* dst,
                        void* stream) {
  TG_CHECK(src && dst && planes > 0 && h > 0 && w > 0 && src != dst, TG_ERR_ARG, "tg_shift: bad args");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(shift_kernel, dim3(grid_for(planes * h * w)), dim3(256), 0, st, src, (long)planes, h, w, dx, dy, dst);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_masked_compose(float* dst, const float* src, const float* mask, int64_t planes, int32_t hw, void* stream) {
  TG_CHECK(dst && src && mask && planes > 0 && hw > 0, TG_ERR_ARG, "tg_masked_compose: bad args");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(compose_kernel, dim3(grid_for(planes * hw)), dim3(256), 0, st, dst, src, mask, (long)planes, hw);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_transpose(int32_t dtype, const void* src, int32_t batch, int32_t rows, int32_t cols, void* dst, void* stream) {
  TG_CHECK((dtype == TG_BF16 || dtype == TG_F16) && src && dst && batch > 0 && rows > 0 && cols > 0 && src != dst, TG_ERR_ARG,
           "tg_transpose: bad args");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (rows % 128 == 0 && cols % 64 == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
    dim3 gridb(cols / 64, rows / 128, batch);
    if (dtype == TG_BF16) hipLaunchKernelGGL(transpose_big_kernel<bf16_t>, gridb, dim3(256), 0, st, (const bf16_t*)src, rows, cols, (bf16_t*)dst);
    else hipLaunchKernelGGL(transpose_big_kernel<f16_t>, gridb, dim3(256), 0, st, (const f16_t*)src, rows, cols, (f16_t*)dst);
    TG_LAUNCH_CHECK();
    return TG_OK;
  }
  dim3 grid((cols + 31) / 32, (rows + 31) / 32, batch);
  if (dtype == TG_BF16) hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)src, rows, cols, (bf16_t*)dst);
  else hipLaunchKernelGGL(transpose_kernel<f16_t>, grid, dim3(256), 0, st, (const f16_t*)src, rows, cols, (f16_t*)dst);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_conv1x1_nchw(const float* x, int32_t batch, int32_t cin, int32_t cout, int64_t hw, const float* weight,
                               const float* bias, float in_scale, float* out, void* stream) {
  TG_CHECK(x && weight && out && batch > 0 && hw > 0 && cin > 0 && cin <= 8 && cout > 0 && cout <= 8 && x != out, TG_ERR_ARG,
           "tg_conv1x1_nchw: bad args (cin, cout <= 8)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long n = (long)batch * hw;
  hipLaunchKernelGGL(conv1x1_nchw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, batch, cin, cout, (long)hw, weight, bias,
                     in_scale, out);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_softmax_rows(int32_t dtype, const void* x, int64_t rows, int32_t cols, int64_t ldx, float scale, void* out,
                               int64_t ldo, void* stream) {
  TG_CHECK((dtype == TG_BF16 || dtype == TG_F16) && x && out && rows > 0 && cols > 0 && cols % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0,
           TG_ERR_ARG, "tg_softmax_rows: bad args (cols and pitches must be multiples of 8)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == TG_BF16) hipLaunchKernelGGL(softmax_rows_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, (long)rows, cols, (long)ldx, scale, (bf16_t*)out, (long)ldo);
  else hipLaunchKernelGGL(softmax_rows_kernel<f16_t>, grid, dim3(256), 0, st, (const f16_t*)x, (long)rows, cols, (long)ldx, scale, (f16_t*)out, (long)ldo);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

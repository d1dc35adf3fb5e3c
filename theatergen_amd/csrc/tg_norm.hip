// GroupNorm (+SiLU) over token-major activations and LayerNorm per token row — HBM-bound kernels.
// 16-byte (8 x bf16/f16) accesses per lane, channels contiguous -> fully coalesced rows.
//
// GroupNorm is two launches: (1) per-slab partial sum / sum-of-squares per (batch, group) — no atomics,
// deterministic; (2) every block folds the partials of its batch item to mean / rstd in fp64 (fixed order), applies
// gamma/beta (+SiLU) and writes the normalised activations once.  The input may be the channel concat of two tensors (skip connection):
// groups may straddle the boundary (e.g. 1280 + 640 channels -> 60 channels per group).
#include "tg_common.h"

namespace {

constexpr int GN_MAX_CHUNKS = 4;  // channel chunks (of 8) per thread column: supports C <= 8 * 256 * 4

struct GnParams {
  const void* x0;
  const void* x1;
  int c0, c1;
  int batch;
  long hw;
  int groups;
  float eps;
  const void* gamma;
  const void* beta;
  int silu;
  void* out;
  float* partials;  // [batch][nblk][groups][2]
  float* stats;     // [batch][groups][2] (mean, rstd) — placed after the partials
  int nblk;
  int npart;        // partial sums per batch item the statistics fold reads (= nblk unless a producer wrote them: tg_groupnorm_from_partials)
  int cx, ry;       // thread grid: cx channel-chunk columns x ry pixel rows
  float* coef;      // tg_groupnorm_coef: [batch][2][C] (a = rstd * gamma, d = beta - mean * a) instead of the normalised tensor
};

template <typename T>
__device__ __forceinline__ typename Vec<T>::v8 gn_load8(const GnParams& p, int b, long pix, int ch) {
  typedef typename Vec<T>::v8 V8;
  if (ch < p.c0) return *reinterpret_cast<const V8*>(reinterpret_cast<const T*>(p.x0) + ((long)b * p.hw + pix) * p.c0 + ch);
  return *reinterpret_cast<const V8*>(reinterpret_cast<const T*>(p.x1) + ((long)b * p.hw + pix) * p.c1 + (ch - p.c0));
}

template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(GnParams p) {
  extern __shared__ float sh[];  // [ry][2][C]: one private row per pixel-row of threads -> no atomics, deterministic
  const int C = p.c0 + p.c1;
  const int cpr = C / 8;
  const int b = blockIdx.y;
  const int blk = blockIdx.x;
  const long per = (p.hw + p.nblk - 1) / p.nblk;
  const long p_begin = (long)blk * per;
  long p_end = p_begin + per;
  if (p_end > p.hw) p_end = p.hw;
  const int tx = threadIdx.x % p.cx;
  const int ty = threadIdx.x / p.cx;
  if (ty < p.ry) {
    float* mine = sh + (size_t)ty * 2 * C;
    for (int c = tx; c < cpr; c += p.cx) {
      float s[8], q[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
      // 4 pixels per trip: four independent 16-byte loads in flight per lane (the kernel is HBM-latency bound otherwise)
      long pix = p_begin + ty;
      for (; pix + 3 * (long)p.ry < p_end; pix += 4 * (long)p.ry) {
        typename Vec<T>::v8 v0 = gn_load8<T>(p, b, pix, c * 8);
        typename Vec<T>::v8 v1 = gn_load8<T>(p, b, pix + p.ry, c * 8);
        typename Vec<T>::v8 v2 = gn_load8<T>(p, b, pix + 2 * (long)p.ry, c * 8);
        typename Vec<T>::v8 v3 = gn_load8<T>(p, b, pix + 3 * (long)p.ry, c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f0 = to_f32<T>(v0[j]), f1 = to_f32<T>(v1[j]), f2 = to_f32<T>(v2[j]), f3 = to_f32<T>(v3[j]);
          s[j] += f0; q[j] += f0 * f0;
          s[j] += f1; q[j] += f1 * f1;
          s[j] += f2; q[j] += f2 * f2;
          s[j] += f3; q[j] += f3 * f3;
        }
      }
      for (; pix < p_end; pix += p.ry) {
        typename Vec<T>::v8 v = gn_load8<T>(p, b, pix, c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float f = to_f32<T>(v[j]);
          s[j] += f;
          q[j] += f * f;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        mine[c * 8 + j] = s[j];
        mine[C + c * 8 + j] = q[j];
      }
    }
  }
  __syncthreads();
  // fold the ry private rows into row 0 in a fixed order
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    float a = sh[i];
    for (int r = 1; r < p.ry; ++r) a += sh[(size_t)r * 2 * C + i];
    sh[i] = a;
  }
  __syncthreads();
  const int cpg = C / p.groups;
  for (int g = threadIdx.x; g < p.groups; g += blockDim.x) {
    float s = 0.f, q = 0.f;
    for (int j = 0; j < cpg; ++j) { s += sh[g * cpg + j]; q += sh[C + g * cpg + j]; }
    float* o = p.partials + (((long)b * p.nblk + blk) * p.groups + g) * 2;
    o[0] = s;
    o[1] = q;
  }
}

// mean / rstd of every (batch b, group) from the slab partials, by ALL 256 threads of the calling block: thread
// (part = t / 32, g = t % 32 + 32 * k) sums every 8th slab partial in fp64, the 8 parts are folded in a fixed order
// through LDS -> deterministic.  Runs as the prologue of the apply kernel (each block re-derives the 32 group
// statistics of its batch item from <= 16 KB of L2-resident partials) instead of as a launch of its own: one launch
// and ~5 us less per GroupNorm, 61 GroupNorms per UNet call.
__device__ __forceinline__ void gn_block_stats(const GnParams& p, int b, float* s_stats /* [groups][2] in LDS */) {
  __shared__ double sh[8][32][2];
  const int C = p.c0 + p.c1;
  const double cnt = (double)p.hw * (C / p.groups);
  const int part = threadIdx.x >> 5, gl = threadIdx.x & 31;
  for (int g0 = 0; g0 < p.groups; g0 += 32) {
    const int g = g0 + gl;
    double s = 0.0, q = 0.0;
    if (g < p.groups)
      for (int k = part; k < p.npart; k += 8) {
        const float* o = p.partials + (((long)b * p.npart + k) * p.groups + g) * 2;
        s += (double)o[0];
        q += (double)o[1];
      }
    sh[part][gl][0] = s;
    sh[part][gl][1] = q;
    __syncthreads();
    if (part == 0 && g < p.groups) {
      for (int r = 1; r < 8; ++r) { s += sh[r][gl][0]; q += sh[r][gl][1]; }
      const double mean = s / cnt;
      double var = q / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      s_stats[g * 2] = (float)mean;
      s_stats[g * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(GnParams p) {
  typedef typename Vec<T>::v8 V8;
  const int C = p.c0 + p.c1;
  const int cpr = C / 8;
  const int cpg = C / p.groups;
  const int b = blockIdx.y;
  const int blk = blockIdx.x;
  const long per = (p.hw + p.nblk - 1) / p.nblk;
  const long p_begin = (long)blk * per;
  long p_end = p_begin + per;
  if (p_end > p.hw) p_end = p.hw;
  extern __shared__ float s_stats[];   // [groups][2]
  const int tx = threadIdx.x % p.cx;
  const int ty = threadIdx.x / p.cx;
  // the first trip's activations do not depend on the statistics: request them before the statistics prologue (a chain
  // of L2 reads and two block barriers) so that their HBM latency runs under it
  V8 pre0, pre1, pre2, pre3;
  const bool have_pre = ty < p.ry && tx < cpr && p_begin + ty + 3 * (long)p.ry < p_end;
  if (have_pre) {
    const long pix = p_begin + ty;
    pre0 = gn_load8<T>(p, b, pix, tx * 8);
    pre1 = gn_load8<T>(p, b, pix + p.ry, tx * 8);
    pre2 = gn_load8<T>(p, b, pix + 2 * (long)p.ry, tx * 8);
    pre3 = gn_load8<T>(p, b, pix + 3 * (long)p.ry, tx * 8);
  }
  gn_block_stats(p, b, s_stats);
  if (ty >= p.ry) return;
  const T* gam = reinterpret_cast<const T*>(p.gamma);
  const T* bet = reinterpret_cast<const T*>(p.beta);
  T* out = reinterpret_cast<T*>(p.out);
  for (int c = tx; c < cpr; c += p.cx) {
    float a[8], d[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = c * 8 + j;
      const int g = ch / cpg;
      const float mean = s_stats[g * 2];
      const float rstd = s_stats[g * 2 + 1];
      const float ga = gam ? to_f32<T>(gam[ch]) : 1.f;
      const float be = bet ? to_f32<T>(bet[ch]) : 0.f;
      a[j] = rstd * ga;
      d[j] = be - mean * a[j];
    }
    auto norm8 = [&](const V8& v) {
      V8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float f = to_f32<T>(v[j]) * a[j] + d[j];
        if (p.silu) f = silu_f(f);
        o[j] = from_f32<T>(f);
      }
      return o;
    };
    // 4 pixels per trip: four independent loads in flight per lane, then four stores
    long pix = p_begin + ty;
    if (c == tx && have_pre) {
      T* o0 = out + ((long)b * p.hw + pix) * C + c * 8;
      *reinterpret_cast<V8*>(o0) = norm8(pre0);
      *reinterpret_cast<V8*>(o0 + (long)p.ry * C) = norm8(pre1);
      *reinterpret_cast<V8*>(o0 + 2 * (long)p.ry * C) = norm8(pre2);
      *reinterpret_cast<V8*>(o0 + 3 * (long)p.ry * C) = norm8(pre3);
      pix += 4 * (long)p.ry;
    }
    for (; pix + 3 * (long)p.ry < p_end; pix += 4 * (long)p.ry) {
      const V8 v0 = gn_load8<T>(p, b, pix, c * 8);
      const V8 v1 = gn_load8<T>(p, b, pix + p.ry, c * 8);
      const V8 v2 = gn_load8<T>(p, b, pix + 2 * (long)p.ry, c * 8);
      const V8 v3 = gn_load8<T>(p, b, pix + 3 * (long)p.ry, c * 8);
      T* o0 = out + ((long)b * p.hw + pix) * C + c * 8;
      *reinterpret_cast<V8*>(o0) = norm8(v0);
      *reinterpret_cast<V8*>(o0 + (long)p.ry * C) = norm8(v1);
      *reinterpret_cast<V8*>(o0 + 2 * (long)p.ry * C) = norm8(v2);
      *reinterpret_cast<V8*>(o0 + 3 * (long)p.ry * C) = norm8(v3);
    }
    for (; pix < p_end; pix += p.ry) {
      const V8 v = gn_load8<T>(p, b, pix, c * 8);
      *reinterpret_cast<V8*>(out + ((long)b * p.hw + pix) * C + c * 8) = norm8(v);
    }
  }
}

// statistics -> per-(batch, channel) coefficients, the prologue of gn_apply_kernel as a launch of its own (one block per
// batch item): a = rstd * gamma, d = beta - mean * a, the expressions gn_apply_kernel evaluates per thread
template <typename T>
__global__ __launch_bounds__(256) void gn_coef_kernel(GnParams p) {
  __shared__ float s_stats[2 * 256];
  const int C = p.c0 + p.c1;
  const int cpg = C / p.groups;
  const int b = blockIdx.x;
  gn_block_stats(p, b, s_stats);
  const T* gam = reinterpret_cast<const T*>(p.gamma);
  const T* bet = reinterpret_cast<const T*>(p.beta);
  float* ca = p.coef + (long)b * 2 * C;
  for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
    const int g = ch / cpg;
    const float mean = s_stats[g * 2];
    const float rstd = s_stats[g * 2 + 1];
    const float ga = gam ? to_f32<T>(gam[ch]) : 1.f;
    const float be = bet ? to_f32<T>(bet[ch]) : 0.f;
    const float a = rstd * ga;
    ca[ch] = a;
    ca[C + ch] = be - mean * a;
  }
}

// SMALL MAPS (hw <= 256: the 16x16 and 8x8 levels) in ONE launch: a block owns one batch item x `gpb` whole groups
// (nc = gpb * cpg / 8 chunk columns of 16 bytes), keeps its whole slab in registers (thread = (chunk column c, pixel lane
// py), MAXP pixels each), and does mean -> centred variance -> normalise (+SiLU) -> store without going back to memory.
// The two-launch path costs >= 11 us on these maps (two launches + the partials round trip) for 3-20 MB of data.
// Needs cpg % 8 == 0 (a 16-byte chunk never straddles groups).  Deterministic: fixed-order LDS folds.
template <typename T, int MAXP>
__global__ __launch_bounds__(256) void gn_small_kernel(GnParams p, int gpb, int nc, int npy) {
  typedef typename Vec<T>::v8 V8;
  __shared__ float red[256];
  __shared__ float col[256];
  __shared__ float gstat[32];
  const int C = p.c0 + p.c1;
  const int cpg = C / p.groups;
  const int ccg = cpg / 8;                     // chunks per group
  const int b = blockIdx.y;
  const int c = threadIdx.x % nc, py = threadIdx.x / nc;
  const bool active = py < npy;
  const int ch0 = blockIdx.x * gpb * cpg + c * 8;
  const int hw = (int)p.hw;
  V8 v[MAXP];
  V8 gv, bv;                                     // gamma / beta of this thread's 8 channels: requested with the slab
  const T* gam = reinterpret_cast<const T*>(p.gamma);
  const T* bet = reinterpret_cast<const T*>(p.beta);
  if (active) {
    if (gam) gv = *reinterpret_cast<const V8*>(gam + ch0);
    if (bet) bv = *reinterpret_cast<const V8*>(bet + ch0);
  }
  float s = 0.f;
  if (active) {
    // every load is issued before the first use (out-of-range pixels re-read the last one and are ignored below): with
    // a branch around each load the compiler serialises load -> wait -> add and the kernel is one latency chain
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int pix = py + i * npy;
      v[i] = gn_load8<T>(p, b, pix < hw ? pix : hw - 1, ch0);
    }
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      if (py + i * npy < hw) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s += to_f32<T>(v[i][j]);
      }
    }
    red[py * nc + c] = s;
  }
  __syncthreads();
  const float inv_n = 1.f / ((float)hw * (float)cpg);
  // fixed-order fold in two short steps (column sums over the pixel lanes, then the ccg columns of a group): one
  // thread per group walking all npy * ccg entries was a serial chain of dependent LDS reads, ~2.5 us per fold
  if ((int)threadIdx.x < nc) {
    float t = 0.f;
    for (int y = 0; y < npy; ++y) t += red[y * nc + threadIdx.x];
    col[threadIdx.x] = t;
  }
  __syncthreads();
  if ((int)threadIdx.x < gpb) {
    float t = 0.f;
    for (int k = 0; k < ccg; ++k) t += col[threadIdx.x * ccg + k];
    gstat[threadIdx.x] = t * inv_n;
  }
  __syncthreads();
  const float mean = active ? gstat[c / ccg] : 0.f;
  float q = 0.f;
  if (active) {
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int pix = py + i * npy;
      if (pix < hw) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = to_f32<T>(v[i][j]) - mean;
          q += d * d;
        }
      }
    }
    red[py * nc + c] = q;
  }
  __syncthreads();
  if ((int)threadIdx.x < nc) {
    float t = 0.f;
    for (int y = 0; y < npy; ++y) t += red[y * nc + threadIdx.x];
    col[threadIdx.x] = t;
  }
  __syncthreads();
  if ((int)threadIdx.x < gpb) {
    float t = 0.f;
    for (int k = 0; k < ccg; ++k) t += col[threadIdx.x * ccg + k];
    gstat[16 + threadIdx.x] = rsqrtf(t * inv_n + p.eps);
  }
  __syncthreads();
  if (!active) return;
  const float rstd = gstat[16 + c / ccg];
  float a[8], d[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float ga = gam ? to_f32<T>(gv[j]) : 1.f;
    const float be = bet ? to_f32<T>(bv[j]) : 0.f;
    a[j] = rstd * ga;
    d[j] = be - mean * a[j];
  }
  if (p.coef != nullptr) {                       // statistics only (the consumer conv applies them while staging its window)
    if (py == 0) {
      float* ca = p.coef + (long)b * 2 * C + ch0;
#pragma unroll
      for (int j = 0; j < 8; ++j) { ca[j] = a[j]; ca[C + j] = d[j]; }
    }
    return;
  }
  T* out = reinterpret_cast<T*>(p.out);
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int pix = py + i * npy;
    if (pix < hw) {
      V8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float f = to_f32<T>(v[i][j]) * a[j] + d[j];
        if (p.silu) f = silu_f(f);
        o[j] = from_f32<T>(f);
      }
      *reinterpret_cast<V8*>(out + ((long)b * hw + pix) * C + ch0) = o;
    }
  }
}

// geometry of the one-launch path; false -> use the two-launch path
bool gn_small_plan(int C, int groups, long hw, int* gpb, int* nc, int* npy, int* maxp) {
  const int cpg = C / groups;
  if (hw > 256 || cpg % 8 != 0) return false;
  int g = 1;
  while (g < groups && (g * cpg < 64 || groups % g != 0)) ++g;       // >= 128 contiguous bytes per pixel per block
  if (groups % g != 0 || g > 16) return false;
  const int n = g * cpg / 8;
  if (n > 256) return false;
  const int y = 256 / n;
  const int mp = (int)((hw + y - 1) / y);
  if (mp > 24) return false;
  *gpb = g; *nc = n; *npy = y; *maxp = mp;
  return true;
}

// pixel slabs per batch item: 512 blocks in all (2 per CU).  More slabs mean more partials for every apply block's
// statistics prologue to fold: 2048 blocks -2 % end to end, 1024 the previous default, 512 +0.6 %, 256 +0.2 %.
int gn_nblk(int batch, long hw) {
  long target = 512 / (batch > 0 ? batch : 1);
  if (target < 1) target = 1;
  long nblk = hw / 8;  // at least 8 pixels per slab
  if (nblk > target) nblk = target;
  if (nblk < 1) nblk = 1;
  return (int)nblk;
}

// LayerNorm: LPR lanes share one token row (64 / LPR rows per wave), each lane owns the 16-byte chunks l, l + LPR, ...
// (MAXC of them) of its row: at C = 320 a row is 40 chunks, so one-row-per-wave left 24 of 64 lanes idle; with LPR = 8
// every lane carries 5 chunks.  Two-pass (mean, then centred sum of squares) in registers, reductions by xor-shuffles
// inside the LPR-lane group.
// stats != NULL: statistics only — (rstd, -rstd * mean) per row, fp32 [rows][2], for the LayerNorm-folded projections (tg_gemm ln_rows):
// the same two-pass mean / centred variance as the normalising kernel, half its traffic (no output tensor).
template <typename T, int LPR, int MAXC>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* x, long rows, int C, long ldx, float eps, const T* gamma,
                                                        const T* beta, T* out, long ldo, float* stats = nullptr) {
  typedef typename Vec<T>::v8 V8;
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int l = lane % LPR;
  const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
  const bool row_ok = row < rows;
  const int cpr = C / 8;
  float v[MAXC][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = l + LPR * i;
    if (row_ok && c < cpr) {
      V8 t = *reinterpret_cast<const V8*>(x + row * ldx + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[i][j] = to_f32<T>(t[j]); s += v[i][j]; }
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = l + LPR * i;
    if (row_ok && c < cpr) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { float dlt = v[i][j] - mean; q += dlt * dlt; }
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / (float)C + eps);
  if (!row_ok) return;
  if (stats != nullptr) {
    if (l == 0) *reinterpret_cast<float2*>(stats + 2 * row) = make_float2(rstd, -rstd * mean);
    return;
  }
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = l + LPR * i;
    if (c < cpr) {
      V8 o;
      V8 g8, b8;
      if (gamma) g8 = *reinterpret_cast<const V8*>(gamma + c * 8);
      if (beta) b8 = *reinterpret_cast<const V8*>(beta + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float f = (v[i][j] - mean) * rstd;
        if (gamma) f *= to_f32<T>(g8[j]);
        if (beta) f += to_f32<T>(b8[j]);
        o[j] = from_f32<T>(f);
      }
      *reinterpret_cast<V8*>(out + row * ldo + c * 8) = o;
    }
  }
}

template <typename T>
int launch_ln(const void* x, long rows, int C, long ldx, float eps, const void* gamma, const void* beta, void* out,
              long ldo, hipStream_t st, float* stats = nullptr) {
  const int cpr = C / 8;
#define LN_CASE(LPR, MAXC)                                                                                              \
  hipLaunchKernelGGL((layernorm_kernel<T, LPR, MAXC>), dim3((unsigned)((rows + 4 * (64 / LPR) - 1) / (4 * (64 / LPR)))), \
                     dim3(256), 0, st, (const T*)x, rows, C, ldx, eps, (const T*)gamma, (const T*)beta, (T*)out, ldo, stats)
  if (cpr <= 8) LN_CASE(8, 1);
  else if (cpr <= 40) LN_CASE(8, 5);          // C <= 320
  else if (cpr <= 80) LN_CASE(16, 5);         // C <= 640
  else if (cpr <= 160) LN_CASE(32, 5);        // C <= 1280
  else if (cpr <= 256) LN_CASE(64, 4);
  else LN_CASE(64, 8);
#undef LN_CASE
  TG_LAUNCH_CHECK();
  return TG_OK;
}

}  // namespace

extern "C" int64_t tg_groupnorm_scratch_bytes(int32_t batch, int64_t hw, int32_t groups) {
  const int nblk = gn_nblk(batch, hw);
  return ((int64_t)batch * nblk * groups * 2 + (int64_t)batch * groups * 2) * 4;
}

namespace {
int groupnorm_impl(int32_t dtype, const void* x0, const void* x1, int32_t c0, int32_t c1, int32_t batch,
                   int64_t hw, int32_t groups, float eps, const void* gamma, const void* beta, int32_t silu,
                   void* out, float* coef, void* partials, void* stream) {
  TG_CHECK(dtype == TG_BF16 || dtype == TG_F16, TG_ERR_ARG, "tg_groupnorm: bad dtype");
  TG_CHECK(x0 && (out || coef) && partials, TG_ERR_ARG, "tg_groupnorm: null pointer");
  if (!x1) c1 = 0;
  const int C = c0 + c1;
  TG_CHECK(batch > 0 && hw > 0 && groups > 0 && C % groups == 0 && c0 % 8 == 0 && c1 % 8 == 0, TG_ERR_ARG,
           "tg_groupnorm: bad shape batch=%d hw=%lld C=%d+%d groups=%d", batch, (long long)hw, c0, c1, groups);
  TG_CHECK(C <= 8 * 256 * GN_MAX_CHUNKS, TG_ERR_ARG, "tg_groupnorm: C too large");
  GnParams p{};
  p.x0 = x0; p.x1 = x1; p.c0 = c0; p.c1 = c1; p.batch = batch; p.hw = hw; p.groups = groups; p.eps = eps;
  p.gamma = gamma; p.beta = beta; p.silu = silu; p.out = out; p.coef = coef;
  TG_CHECK(coef == nullptr || groups <= 256, TG_ERR_ARG, "tg_groupnorm_coef: groups <= 256");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  {
    int gpb, nc, npy, maxp;
    const bool vec_ok = reinterpret_cast<uintptr_t>(gamma) % 16 == 0 && reinterpret_cast<uintptr_t>(beta) % 16 == 0;
    if (vec_ok && gn_small_plan(C, groups, hw, &gpb, &nc, &npy, &maxp)) {
      dim3 grid(groups / gpb, batch);
#define TG_GN_SMALL(MP)                                                                                             \
  do {                                                                                                              \
    if (dtype == TG_BF16) hipLaunchKernelGGL((gn_small_kernel<bf16_t, MP>), grid, dim3(256), 0, st, p, gpb, nc, npy); \
    else hipLaunchKernelGGL((gn_small_kernel<f16_t, MP>), grid, dim3(256), 0, st, p, gpb, nc, npy);                   \
  } while (0)
      if (maxp <= 3) TG_GN_SMALL(3);
      else if (maxp <= 6) TG_GN_SMALL(6);
      else if (maxp <= 12) TG_GN_SMALL(12);
      else TG_GN_SMALL(24);
#undef TG_GN_SMALL
      TG_LAUNCH_CHECK();
      return TG_OK;
    }
  }
  p.nblk = gn_nblk(batch, hw);
  p.npart = p.nblk;
  p.partials = reinterpret_cast<float*>(partials);
  p.stats = p.partials + (long)batch * p.nblk * groups * 2;
  const int cpr = C / 8;
  p.cx = cpr < 256 ? cpr : 256;
  p.ry = 256 / p.cx;
  dim3 grid(p.nblk, batch);
  const size_t lds = (size_t)p.ry * 2 * C * sizeof(float);
  if (dtype == TG_BF16) hipLaunchKernelGGL(gn_partial_kernel<bf16_t>, grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL(gn_partial_kernel<f16_t>, grid, dim3(256), lds, st, p);
  TG_LAUNCH_CHECK();
  if (coef != nullptr) {
    if (dtype == TG_BF16) hipLaunchKernelGGL(gn_coef_kernel<bf16_t>, dim3(batch), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(gn_coef_kernel<f16_t>, dim3(batch), dim3(256), 0, st, p);
    TG_LAUNCH_CHECK();
    return TG_OK;
  }
  const size_t lds_stats = (size_t)groups * 2 * sizeof(float);
  if (dtype == TG_BF16) hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, grid, dim3(256), lds_stats, st, p);
  else hipLaunchKernelGGL(gn_apply_kernel<f16_t>, grid, dim3(256), lds_stats, st, p);
  TG_LAUNCH_CHECK();
  return TG_OK;
}
}  // namespace

extern "C" int tg_groupnorm(int32_t dtype, const void* x0, const void* x1, int32_t c0, int32_t c1, int32_t batch,
                            int64_t hw, int32_t groups, float eps, const void* gamma, const void* beta, int32_t silu,
                            void* out, void* partials, void* stream) {
  TG_CHECK(out != nullptr, TG_ERR_ARG, "tg_groupnorm: null out");
  return groupnorm_impl(dtype, x0, x1, c0, c1, batch, hw, groups, eps, gamma, beta, silu, out, nullptr, partials, stream);
}

extern "C" int tg_groupnorm_coef(int32_t dtype, const void* x0, const void* x1, int32_t c0, int32_t c1, int32_t batch,
                                 int64_t hw, int32_t groups, float eps, const void* gamma, const void* beta, float* coef,
                                 void* partials, void* stream) {
  TG_CHECK(coef != nullptr, TG_ERR_ARG, "tg_groupnorm_coef: null coef");
  return groupnorm_impl(dtype, x0, x1, c0, c1, batch, hw, groups, eps, gamma, beta, 0, nullptr, coef, partials, stream);
}

extern "C" int tg_layernorm(int32_t dtype, const void* x, int64_t rows, int32_t C, int64_t ldx, float eps,
                            const void* gamma, const void* beta, void* out, int64_t ldo, void* stream) {
  TG_CHECK(dtype == TG_BF16 || dtype == TG_F16, TG_ERR_ARG, "tg_layernorm: bad dtype");
  TG_CHECK(x && out && rows > 0 && C > 0 && C % 8 == 0 && C <= 8 * 64 * 8 && ldx % 8 == 0 && ldo % 8 == 0, TG_ERR_ARG,
           "tg_layernorm: bad args rows=%lld C=%d", (long long)rows, C);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == TG_BF16) return launch_ln<bf16_t>(x, rows, C, ldx, eps, gamma, beta, out, ldo, st);
  return launch_ln<f16_t>(x, rows, C, ldx, eps, gamma, beta, out, ldo, st);
}

extern "C" int tg_groupnorm_from_partials(int32_t dtype, const void* x, int32_t C, int32_t batch, int64_t hw, int32_t groups, float eps, const void* gamma,
                                          const void* beta, int32_t silu, void* out, float* coef, const float* partials, int32_t nblk, void* stream) {
  TG_CHECK(dtype == TG_BF16 || dtype == TG_F16, TG_ERR_ARG, "tg_groupnorm_from_partials: bad dtype");
  TG_CHECK(partials != nullptr && nblk > 0 && (coef != nullptr || (x != nullptr && out != nullptr)), TG_ERR_ARG, "tg_groupnorm_from_partials: null pointer");
  TG_CHECK(batch > 0 && hw > 0 && groups > 0 && groups <= 256 && C % groups == 0 && C % 8 == 0 && C <= 8 * 256 * GN_MAX_CHUNKS, TG_ERR_ARG,
           "tg_groupnorm_from_partials: bad shape batch=%d hw=%lld C=%d groups=%d", batch, (long long)hw, C, groups);
  GnParams p{};
  p.x0 = x; p.x1 = nullptr; p.c0 = C; p.c1 = 0; p.batch = batch; p.hw = hw; p.groups = groups; p.eps = eps;
  p.gamma = gamma; p.beta = beta; p.silu = silu; p.out = out; p.coef = coef;
  p.partials = const_cast<float*>(partials);
  p.npart = nblk;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (coef != nullptr) {
    if (dtype == TG_BF16) hipLaunchKernelGGL(gn_coef_kernel<bf16_t>, dim3(batch), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(gn_coef_kernel<f16_t>, dim3(batch), dim3(256), 0, st, p);
    TG_LAUNCH_CHECK();
    return TG_OK;
  }
  p.nblk = gn_nblk(batch, hw);                     // pixel slabs of the apply pass (independent of the producer's block count)
  const int cpr = C / 8;
  p.cx = cpr < 256 ? cpr : 256;
  p.ry = 256 / p.cx;
  const size_t lds_stats = (size_t)groups * 2 * sizeof(float);
  if (dtype == TG_BF16) hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, dim3(p.nblk, batch), dim3(256), lds_stats, st, p);
  else hipLaunchKernelGGL(gn_apply_kernel<f16_t>, dim3(p.nblk, batch), dim3(256), lds_stats, st, p);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_layernorm_stats(int32_t dtype, const void* x, int64_t rows, int32_t C, int64_t ldx, float eps, float* stats, void* stream) {
  TG_CHECK(dtype == TG_BF16 || dtype == TG_F16, TG_ERR_ARG, "tg_layernorm_stats: bad dtype");
  TG_CHECK(x && stats && rows > 0 && C > 0 && C % 8 == 0 && C <= 8 * 64 * 8 && ldx % 8 == 0 && (reinterpret_cast<uintptr_t>(stats) & 7) == 0, TG_ERR_ARG,
           "tg_layernorm_stats: bad args rows=%lld C=%d", (long long)rows, C);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == TG_BF16) return launch_ln<bf16_t>(x, rows, C, ldx, eps, nullptr, nullptr, nullptr, 0, st, stats);
  return launch_ln<f16_t>(x, rows, C, ldx, eps, nullptr, nullptr, nullptr, 0, st, stats);
}

// MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (see include/theatergen_hip.h: tg_gemm).
//
// out[m, n] = epilogue( sum_k A[m, k] * W[n, k] ),  A token-major activations, W [N, K] (K contiguous).
// The MFMA is issued "swapped": A-operand = W rows (n), B-operand = activation rows (m), so the
// 32x32 accumulator tile has lane&31 = token and 4 consecutive registers = 4 consecutive output
// channels -> 8-byte stores / 8-byte bias+residual loads in the epilogue, and per-token quantities
// stay lane-local.
//
// Block tile BM(tokens) x BN(channels) x BK=64, 256 threads = 4 waves (WAVES_M x WAVES_N), each wave
// TM x TN tiles of v_mfma_f32_32x32x16.  Operands are register-staged (global_load_dwordx4 issued
// before the MFMA block of the current tile, ds_write_b128 after it) into double-buffered LDS with a
// 144-byte row pitch (128 B of K + 16 B pad): ds_read_b128 of 16 rows x one 16-B column hits 16 distinct
// 4-bank slots (conflict-free), ds_write_b128 rows are contiguous.
//
// Conv mode gathers the A rows on the fly (no im2col buffer): K is tap-major (ky, kx, c); a BK chunk
// never straddles a tap because channel counts are multiples of 64, so each A row of a K-tile is one
// contiguous 128-B segment of a shifted input pixel (or zeros at the border).  Stride-2 (Downsample2D),
// nearest-x2 upsampled input (Upsample2D) and a two-source channel concat (skip connection) are folded
// into the gather.
#include "tg_common.h"

namespace {

constexpr int BK = 64;
constexpr int LDP = BK + 8;  // LDS row pitch in elements (144 B)

struct GemmParams {
  const void* a0;
  const void* a1;
  int c0, c1;
  int in_h, in_w, out_h, out_w, stride, upsample;
  const void* w;
  long M, N, K;
  const void* bias;
  const void* bvec;
  long ldbvec;
  long rows_per_batch;
  const void* res;
  long ldres;
  int act;
  float out_scale;
  void* out;
  long ldc;
  long n_split;
  void* out_t;
  long ldt;
  float* ws;
  int splits;
  int kt_per_split;
  int tiles_n;
  long a_rpb, a_bs;
};

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
  if (p.act == TG_ACT_SILU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = silu_f(v[j]);
  } else if (p.act == TG_ACT_GELU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = gelu_erf_f(v[j]);
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

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool CONV>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  constexpr int TM = BM / (WAVES_M * 32);
  constexpr int TN = BN / (WAVES_N * 32);
  constexpr int XR = BM / 32;  // 16-B loads per thread for the activation tile
  constexpr int WR = BN / 32;
  typedef typename Vec<T>::v8 V8;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sX = reinterpret_cast<T*>(smem);                 // [2][BM][LDP]
  T* sW = sX + 2 * BM * LDP;                          // [2][BN][LDP]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_m = wave / WAVES_N;
  const int wave_n = wave % WAVES_N;
  const int tile_n = blockIdx.x % p.tiles_n;
  const int tile_m = blockIdx.x / p.tiles_n;
  const long m0 = (long)tile_m * BM;
  const long n0 = (long)tile_n * BN;
  const int split = blockIdx.z;
  const int nkt_total = (int)((p.K + BK - 1) / BK);
  const int kt_begin = split * p.kt_per_split;
  int kt_end = kt_begin + p.kt_per_split;
  if (kt_end > nkt_total) kt_end = nkt_total;
  const int nkt = kt_end - kt_begin;

  const int chunk = tid & 7;
  const int lrow = tid >> 3;  // 0..31

  const T* A0 = reinterpret_cast<const T*>(p.a0);
  const T* A1 = reinterpret_cast<const T*>(p.a1);
  const T* Wp = reinterpret_cast<const T*>(p.w);
  const int ctot = p.c0 + p.c1;

  // per-thread row bookkeeping for the activation gather
  long xbase[XR];   // plain: element offset of the row in source 0
  long xrow[XR];
  int x_oy[XR], x_ox[XR], x_ob[XR];
  bool x_ok[XR];
#pragma unroll
  for (int i = 0; i < XR; ++i) {
    long m = m0 + lrow + 32 * i;
    x_ok[i] = m < p.M;
    xbase[i] = m * p.c0;
    xrow[i] = m;
    if (!CONV && p.a_rpb > 0) { const long bb = m / p.a_rpb; xbase[i] = bb * p.a_bs + (m - bb * p.a_rpb) * p.c0; }
    if (CONV) {
      long mm = x_ok[i] ? m : 0;
      int hw = p.out_h * p.out_w;
      x_ob[i] = (int)(mm / hw);
      int r = (int)(mm - (long)x_ob[i] * hw);
      x_oy[i] = r / p.out_w;
      x_ox[i] = r - x_oy[i] * p.out_w;
    }
  }

  u32x4 xreg[XR], wreg[WR];

  auto load_tile = [&](int kt) {
    const long k0 = (long)kt * BK;
    const long kc = k0 + chunk * 8;
    // ---- weights
    {
      const bool kok = kc < p.K;
#pragma unroll
      for (int i = 0; i < WR; ++i) {
        long n = n0 + lrow + 32 * i;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (kok && n < p.N) v = *reinterpret_cast<const u32x4*>(Wp + n * p.K + kc);
        wreg[i] = v;
      }
    }
    // ---- activations
    if (!CONV) {
      const T* src = A0;
      long pitch = p.c0;
      long kk = kc;
      bool kok = kc < p.K;
      const bool second = A1 != nullptr && k0 >= p.c0;
      if (second) { src = A1; pitch = p.c1; kk = kc - p.c0; }
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        u32x4 v = {0u, 0u, 0u, 0u};
        const long off = second ? xrow[i] * pitch : xbase[i];
        if (kok && x_ok[i]) v = *reinterpret_cast<const u32x4*>(src + off + kk);
        xreg[i] = v;
      }
    } else {
      const int tap = (int)(k0 / ctot);
      int cc = (int)(k0 - (long)tap * ctot);
      const int ky = tap / 3, kx = tap - ky * 3;
      const T* src = A0;
      int pitch = p.c0;
      if (cc >= p.c0) { src = A1; pitch = p.c1; cc -= p.c0; }
      cc += chunk * 8;
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        int iy, ix;
        bool ok = x_ok[i];
        if (!p.upsample) {
          iy = x_oy[i] * p.stride + ky - 1;
          ix = x_ox[i] * p.stride + kx - 1;
          ok = ok && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
        } else {
          int uy = x_oy[i] + ky - 1, ux = x_ox[i] + kx - 1;
          ok = ok && uy >= 0 && uy < 2 * p.in_h && ux >= 0 && ux < 2 * p.in_w;
          iy = uy >> 1;
          ix = ux >> 1;
        }
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ok) v = *reinterpret_cast<const u32x4*>(src + ((long)(x_ob[i] * p.in_h + iy) * p.in_w + ix) * pitch + cc);
        xreg[i] = v;
      }
    }
  };

  auto store_tile = [&](int buf) {
    T* dx = sX + buf * BM * LDP;
    T* dw = sW + buf * BN * LDP;
#pragma unroll
    for (int i = 0; i < XR; ++i) *reinterpret_cast<u32x4*>(dx + (lrow + 32 * i) * LDP + chunk * 8) = xreg[i];
#pragma unroll
    for (int i = 0; i < WR; ++i) *reinterpret_cast<u32x4*>(dw + (lrow + 32 * i) * LDP + chunk * 8) = wreg[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nkt > 0) {
    load_tile(kt_begin);
    store_tile(0);
    __syncthreads();
    const int frow = lane & 31;
    const int fk = (lane >> 5) * 8;
    for (int it = 0; it < nkt; ++it) {
      const int buf = it & 1;
      if (it + 1 < nkt) load_tile(kt_begin + it + 1);
      const T* bx = sX + buf * BM * LDP + (wave_m * TM * 32 + frow) * LDP + fk;
      const T* bw = sW + buf * BN * LDP + (wave_n * TN * 32 + frow) * LDP + fk;
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        V8 xf[TM], wf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) xf[i] = *reinterpret_cast<const V8*>(bx + i * 32 * LDP + ks * 16);
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const V8*>(bw + j * 32 * LDP + ks * 16);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(wf[j], xf[i], acc[i][j]);
      }
      if (it + 1 < nkt) store_tile(buf ^ 1);
      __syncthreads();
    }
  }

  // ---- epilogue: lane&31 = token, regs = channels
  const int hi = lane >> 5;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long m = m0 + (wave_m * TM + i) * 32 + (lane & 31);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const long nb = n0 + (wave_n * TN + j) * 32 + 4 * hi;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const long n4 = nb + 8 * g;
        if (p.splits > 1) {
          if (m < p.M && n4 < p.N) {
            f32x4 o = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            *reinterpret_cast<f32x4*>(p.ws + ((long)split * p.M + m) * p.N + n4) = o;
          }
        } else {
          epilogue_store4<T>(p, m, n4, acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p) {
  const long n4s = p.N / 4;
  const long total = p.M * n4s;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long m = idx / n4s;
    const long n4 = (idx - m * n4s) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < p.splits; ++z) s += *reinterpret_cast<const f32x4*>(p.ws + ((long)z * p.M + m) * p.N + n4);
    epilogue_store4<T>(p, m, n4, s[0], s[1], s[2], s[3]);
  }
}

struct TileCfg { int bm, bn; };
const TileCfg kTiles[] = {{128, 128}, {64, 64}, {128, 64}, {64, 128}};
constexpr int kNumTiles = 4;

struct Plan { int tile; int splits; int kt_per_split; long tiles_m, tiles_n; };

Plan make_plan(const tg_gemm_desc* d) {
  const long M = d->M, N = d->N, K = d->K;
  const int nkt = (int)((K + BK - 1) / BK);
  Plan best{};
  double best_cost = 1e300;
  for (int t = 0; t < kNumTiles; ++t) {
    if (d->force_tile > 0 && d->force_tile - 1 != t) continue;
    const long tm = (M + kTiles[t].bm - 1) / kTiles[t].bm, tn = (N + kTiles[t].bn - 1) / kTiles[t].bn;
    const long tiles = tm * tn;
    const int lds = 2 * (kTiles[t].bm + kTiles[t].bn) * LDP * 2;
    const int wg_per_cu = lds <= 40 * 1024 ? 4 : (lds <= 53 * 1024 ? 3 : 2);
    const long slots = 256L * wg_per_cu;
    for (int s = 1; s <= 32; s = (s < 4 ? s + 1 : s * 2)) {
      if (d->force_split_k > 0 && s != d->force_split_k) continue;
      if (d->force_split_k <= 0 && s > 1 && nkt / s < 6) break;
      const int kps = (nkt + s - 1) / s;
      const int real_s = (nkt + kps - 1) / kps;
      if (real_s != s) continue;
      const long wgs = tiles * s;
      const double rounds = (double)((wgs + slots - 1) / slots);
      // relative MFMA efficiency of a tile config (bigger tiles amortise LDS traffic better)
      const double eff = (t == 0) ? 1.0 : (t == 1 ? 0.62 : 0.8);
      double cost = rounds * wg_per_cu * (double)kTiles[t].bm * kTiles[t].bn * (kps * BK + 96) / eff;
      if (s > 1) cost += 2.5 * (double)M * N * s * 4.0 / 256.0 * 6.0;  // fp32 partial write+read
      if (cost < best_cost) { best_cost = cost; best = Plan{t, s, kps, tm, tn}; }
    }
  }
  return best;
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_cfg(const tg_gemm_desc* d, const GemmParams& p, const Plan& pl, hipStream_t st) {
  const size_t lds = (size_t)2 * (BM + BN) * LDP * sizeof(T);
  dim3 grid((unsigned)(pl.tiles_m * pl.tiles_n), 1, (unsigned)pl.splits);
  if (d->mode == 1) {
    auto k = gemm_kernel<T, BM, BN, WM, WN, true>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)attr;
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
  } else {
    auto k = gemm_kernel<T, BM, BN, WM, WN, false>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)attr;
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
  }
  TG_LAUNCH_CHECK();
  if (pl.splits > 1) {
    long total = p.M * (p.N / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3(blocks), dim3(256), 0, st, p);
    TG_LAUNCH_CHECK();
  }
  return TG_OK;
}

template <typename T>
int launch_gemm(const tg_gemm_desc* d, hipStream_t st) {
  Plan pl = make_plan(d);
  GemmParams p{};
  p.a0 = d->a0; p.a1 = d->a1; p.c0 = d->c0; p.c1 = d->a1 ? d->c1 : 0;
  p.in_h = d->in_h; p.in_w = d->in_w; p.out_h = d->out_h; p.out_w = d->out_w;
  p.stride = d->stride; p.upsample = d->upsample;
  p.w = d->w; p.M = d->M; p.N = d->N; p.K = d->K;
  p.bias = d->bias; p.bvec = d->bvec; p.ldbvec = d->ldbvec;
  p.rows_per_batch = d->rows_per_batch > 0 ? d->rows_per_batch : d->M;
  p.res = d->res; p.ldres = d->ldres; p.act = d->act; p.out_scale = d->out_scale;
  p.out = d->out; p.ldc = d->ldc; p.n_split = d->n_split; p.out_t = d->out_t; p.ldt = d->ldt;
  p.ws = reinterpret_cast<float*>(d->workspace);
  p.splits = pl.splits; p.kt_per_split = pl.kt_per_split; p.tiles_n = (int)pl.tiles_n;
  p.a_rpb = d->mode == 0 ? d->a_rows_per_batch : 0; p.a_bs = d->a_batch_stride;
  if (pl.splits > 1) {
    TG_CHECK(d->workspace != nullptr && d->workspace_bytes >= (int64_t)pl.splits * d->M * d->N * 4, TG_ERR_ARG,
             "tg_gemm: split-K needs %lld workspace bytes, got %lld", (long long)pl.splits * d->M * d->N * 4,
             (long long)d->workspace_bytes);
  }
  switch (pl.tile) {
    case 0: return launch_cfg<T, 128, 128, 2, 2>(d, p, pl, st);
    case 1: return launch_cfg<T, 64, 64, 2, 2>(d, p, pl, st);
    case 2: return launch_cfg<T, 128, 64, 4, 1>(d, p, pl, st);
    default: return launch_cfg<T, 64, 128, 1, 4>(d, p, pl, st);
  }
}

int validate(const tg_gemm_desc* d) {
  TG_CHECK(d != nullptr, TG_ERR_ARG, "tg_gemm: null descriptor");
  TG_CHECK(d->dtype == TG_BF16 || d->dtype == TG_F16, TG_ERR_ARG, "tg_gemm: bad dtype %d", d->dtype);
  TG_CHECK(d->a0 && d->w && d->out, TG_ERR_ARG, "tg_gemm: null a0/w/out");
  TG_CHECK(d->M > 0 && d->N > 0 && d->K > 0, TG_ERR_ARG, "tg_gemm: empty problem M=%lld N=%lld K=%lld",
           (long long)d->M, (long long)d->N, (long long)d->K);
  TG_CHECK(d->N % 4 == 0 && d->K % 8 == 0, TG_ERR_ARG, "tg_gemm: N %% 4 and K %% 8 required (N=%lld K=%lld)",
           (long long)d->N, (long long)d->K);
  TG_CHECK(d->geglu == 0, TG_ERR_UNSUPPORTED, "tg_gemm: fused GEGLU epilogue not available; use tg_geglu");
  const int ctot = d->c0 + (d->a1 ? d->c1 : 0);
  if (d->a1) TG_CHECK(d->c0 % BK == 0, TG_ERR_ARG, "tg_gemm: two-source A needs c0 %% 64 == 0 (c0=%d)", d->c0);
  if (d->mode == 1) {
    TG_CHECK(ctot % BK == 0, TG_ERR_ARG, "tg_gemm conv: channels %% 64 required (c=%d)", ctot);
    TG_CHECK(d->K == 9L * ctot, TG_ERR_ARG, "tg_gemm conv: K must be 9*(c0+c1)");
    TG_CHECK(d->stride == 1 || d->stride == 2, TG_ERR_ARG, "tg_gemm conv: stride 1|2");
    TG_CHECK(!(d->upsample && d->stride != 1), TG_ERR_ARG, "tg_gemm conv: upsample needs stride 1");
    TG_CHECK(d->M == (int64_t)d->batch * d->out_h * d->out_w, TG_ERR_ARG, "tg_gemm conv: M != batch*out_h*out_w");
    const int eh = d->upsample ? 2 * d->in_h : (d->in_h + 2 - 3) / d->stride + 1;
    const int ew = d->upsample ? 2 * d->in_w : (d->in_w + 2 - 3) / d->stride + 1;
    TG_CHECK(eh == d->out_h && ew == d->out_w, TG_ERR_ARG, "tg_gemm conv: out %dx%d inconsistent with in %dx%d",
             d->out_h, d->out_w, d->in_h, d->in_w);
  } else {
    TG_CHECK(d->mode == 0, TG_ERR_ARG, "tg_gemm: bad mode %d", d->mode);
    TG_CHECK(ctot == d->K, TG_ERR_ARG, "tg_gemm: K (%lld) != c0+c1 (%d)", (long long)d->K, ctot);
    if (d->a_rows_per_batch > 0)
      TG_CHECK(d->a1 == nullptr && d->a_batch_stride % 8 == 0, TG_ERR_ARG, "tg_gemm: batched A needs a single source and a 16-byte aligned batch pitch");
  }
  if (d->n_split > 0) {
    TG_CHECK(d->out_t && d->n_split % 4 == 0 && d->rows_per_batch > 0, TG_ERR_ARG, "tg_gemm: bad transposed-output args");
  }
  if (d->bvec) TG_CHECK(d->rows_per_batch > 0, TG_ERR_ARG, "tg_gemm: bvec needs rows_per_batch");
  return TG_OK;
}

}  // namespace

extern "C" int64_t tg_gemm_workspace_bytes(const tg_gemm_desc* d) {
  if (validate(d) != TG_OK) return -1;
  Plan pl = make_plan(d);
  return pl.splits > 1 ? (int64_t)pl.splits * d->M * d->N * 4 : 0;
}

extern "C" int tg_gemm(const tg_gemm_desc* d, void* stream) {
  int rc = validate(d);
  if (rc != TG_OK) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == TG_BF16) return launch_gemm<bf16_t>(d, st);
  return launch_gemm<f16_t>(d, st);
}

// Skinny projections (round 4): out[m, :] = act([LayerNorm-folded] x[m, :] W^T + bias) + res[m, :] for a HANDFUL of rows (M <= a few times 32) — the
// latent path of the Perceiver `Resampler` (ip_adapter/resampler.py:34-78 `PerceiverAttention.to_q / to_kv / to_out`, :13-20 `FeedForward`, :94-97 `proj_out`:
// 16 latents x (cond, zero-image) = 32 rows against 1280 x 1280 ... 5120 x 1280 weights) and any other M <= 32 linear (TimestepEmbedding).
//
// Why its own kernel: with 32 rows the LDS-tiled GEMM has N / 128 workgroups (10 ... 40 of 256 CUs), each walking the whole K serially through an LDS pipeline built
// for operand REUSE — there is none here, every weight element is used once.  The Resampler was ~48 dependent launches of 8 ... 12 us each (505 us graph-replayed for
// 166 MB of weights = 21 us of HBM time).  Here:
//   * the 32 rows are the MFMA B operand and live in registers (lane & 31 = row, lane >> 5 = which 8 of a k-step's 16 columns);
//   * the weights are the A operand, packed once on the host in fragment order (weights_pack.skinny_pack: 1 KiB per (32 output rows, 16 k) block, lane-linear), and
//     go global -> registers directly in whole-KiB loads, all of a wave's loads of a chunk in flight at once — no LDS, no barrier in the stream;
//   * a workgroup = one 32-column output tile, its NW = 8 (K % 128 == 0) or 4 waves split K in contiguous slices (N = 1280 -> 40 workgroups x 8 waves, each streaming
//     K / 8 x 32 x 2 bytes: the loads in flight per CU, not the CU count, bound a launch this small), partial accumulators meet in LDS, and the epilogue (fold, bias,
//     GELU, residual, ROUTING) runs on all threads;
//   * output ROUTING: up to three column segments, each plain or transposed with a per-batch-item stride — one launch writes q, the latents' K rows behind the image
//     tokens' rows of the attention's K buffer, and their V^T columns (resampler.py:63-68: `kv_input = cat(x, latents)`), so no concat and no copy is needed.
// LayerNorm fold (single chunk, K <= 1280): statistics two-pass from the register rows (mean, then centred sum of squares), y = rstd (x W'^T - mean u) + v as in
// tg_gemm_glds.h.  One rounding at the end, fp32 before: the rounding points of the tg_gemm path it replaces.
#include "tg_common.h"

namespace {

struct SkinnySeg {
  void* ptr;        // segment base (already offset to the first row / column this launch writes)
  long ld;          // plain: row pitch; transposed: pitch of an output COLUMN's row
  long bs;          // elements between batch items
  int n_end;        // columns [previous n_end, n_end)
  int transposed;
};

struct SkinnyParams {
  const void* x;
  long ldx;
  const void* wpk;
  const float* u;       // fold: row sums of the rounded W gamma
  const float* v;       // fold: W beta (+ bias), fp32
  const void* bias;     // storage dtype [N] or NULL
  const void* res;
  long ldres;
  SkinnySeg seg[3];
  int nseg;
  int M, N, K;
  int rows_per_batch;
  int act;
  int ln;
  float ln_eps;
};

template <typename V>
__device__ __forceinline__ V sk_gld(const void* ptr) { return *(const __attribute__((address_space(1))) V*)ptr; }

// NW waves; CK: k-steps (of 16) per wave and chunk; a wave's K slice is K / (16 NW) k-steps = a whole number of chunks
template <typename T, int NW, int CK>
__global__ __launch_bounds__(NW * 64) void skinny_gemm_kernel(SkinnyParams p) {
  typedef typename Vec<T>::v8 V8;
  __shared__ float accs[NW][16][64];
  __shared__ float red[2][NW][32];
  __shared__ float s_mean[32], s_rstd[32];

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = blockIdx.x, m0 = blockIdx.y * 32;
  const int ksw = p.K / (16 * NW);                   // k-steps per wave
  long row = m0 + l31;
  if (row >= p.M) row = p.M - 1;
  const T* xr = reinterpret_cast<const T*>(p.x) + row * p.ldx + (long)wave * ksw * 16 + 8 * hi;
  const T* wp = reinterpret_cast<const T*>(p.wpk) + (((long)nt * (p.K >> 4) + (long)wave * ksw) * 64 + lane) * 8;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // AH: the weight fragments of a chunk are fetched in AH pieces of AC k-steps (round 5): with CK = 16 / 20 the 2 CK fragment registers + their
  // 64-bit addresses did not fit the 256 registers of an 8-wave workgroup (44 .. 164 bytes of scratch); the rows (b) stay whole for the LayerNorm fold
  constexpr int AH = CK > 12 ? 2 : 1, AC = CK / AH;
  for (int c = 0; c < ksw; c += CK) {
    V8 b[CK];
#pragma unroll
    for (int s = 0; s < CK; ++s) b[s] = sk_gld<V8>(xr + (c + s) * 16);
    V8 a[AC];
#pragma unroll
    for (int s = 0; s < AC; ++s) a[s] = sk_gld<V8>(wp + (long)(c + s) * 512);
    if (p.ln) {
      // (single chunk: ksw == CK, checked by the host) row statistics from the registers, two-pass like tg_layernorm
      float sum = 0.f;
#pragma unroll
      for (int s = 0; s < CK; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += to_f32<T>(b[s][e]);
      sum += __shfl_xor(sum, 32, 64);
      if (hi == 0) red[0][wave][l31] = sum;
      __syncthreads();
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) tot += red[0][w][l31];
      const float mean = tot / (float)p.K;
      float c2 = 0.f;
      // opaque to the optimiser: without it hipcc keeps the fp32 conversions of the first pass (8 CK registers) alive for the second one
#pragma unroll
      for (int s = 0; s < CK; ++s) asm volatile("" : "+v"(b[s]));
#pragma unroll
      for (int s = 0; s < CK; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = to_f32<T>(b[s][e]) - mean; c2 = __builtin_fmaf(d, d, c2); }
      c2 += __shfl_xor(c2, 32, 64);
      if (hi == 0) red[1][wave][l31] = c2;
      __syncthreads();
      if (tid < 32) {
        float q = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) q += red[1][w][tid];
        const float var = q / (float)p.K;
        s_mean[tid] = mean;
        s_rstd[tid] = 1.0f / sqrtf(var + p.ln_eps);
      }
    }
#pragma unroll
    for (int h = 0; h < AH; ++h) {
#pragma unroll
      for (int s = 0; s < AC; ++s) acc = mfma32(a[s], b[h * AC + s], acc);
      if (h + 1 < AH) {
#pragma unroll
        for (int s = 0; s < AC; ++s) a[s] = sk_gld<V8>(wp + (long)(c + (h + 1) * AC + s) * 512);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) accs[wave][r][lane] = acc[r];
  __syncthreads();

  // epilogue on all 256 threads: element e = (row m = e / 32, column nl = e % 32); accumulator register r of lane (m, hi) holds column 8 (r >> 2) + 4 hi + (r & 3)
  const int n_tile = nt * 32;
  int si = 0;
  while (si + 1 < p.nseg && n_tile >= p.seg[si].n_end) ++si;
  const SkinnySeg sg = p.seg[si];
  const int n_seg0 = si == 0 ? 0 : p.seg[si - 1].n_end;
  const T* biasp = reinterpret_cast<const T*>(p.bias);
  const T* resp = reinterpret_cast<const T*>(p.res);
#pragma unroll
  for (int i = 0; i < 16 / NW; ++i) {
    const int e = tid + NW * 64 * i;
    const int ml = e >> 5, nl = e & 31;
    const int m = m0 + ml;
    if (m >= p.M) continue;
    const int h = (nl >> 2) & 1, r = ((nl >> 3) << 2) | (nl & 3);
    float val = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) val += accs[w][r][ml + 32 * h];
    const int n = n_tile + nl;
    if (p.ln) val = s_rstd[ml] * (val - s_mean[ml] * p.u[n]) + p.v[n];
    if (biasp) val += to_f32<T>(biasp[n]);
    if (resp) val += to_f32<T>(resp[(long)m * p.ldres + n]);
    val = apply_act(val, p.act);
    const int bi = m / p.rows_per_batch, rr = m - bi * p.rows_per_batch;
    const int ns = n - n_seg0;
    T* op = reinterpret_cast<T*>(sg.ptr) + (long)bi * sg.bs + (sg.transposed ? (long)ns * sg.ld + rr : (long)rr * sg.ld + ns);
    *op = from_f32<T>(val);
  }
}

template <typename T, int NW>
int launch_skinny_nw(const SkinnyParams& p, hipStream_t st) {
  const int ksw = p.K / (16 * NW);
  dim3 grid((unsigned)(p.N / 32), (unsigned)((p.M + 31) / 32));
#define TG_SK(CKV) hipLaunchKernelGGL((skinny_gemm_kernel<T, NW, CKV>), grid, dim3(NW * 64), 0, st, p)
  // chunks above 12 k-steps only where the LayerNorm fold needs the whole row slice in registers (their weight fragments come in two pieces)
  if (p.ln && ksw % 20 == 0) TG_SK(20);
  else if (p.ln && ksw % 16 == 0) TG_SK(16);
  else if (ksw % 12 == 0) TG_SK(12);
  else if (ksw % 10 == 0) TG_SK(10);
  else if (ksw % 8 == 0) TG_SK(8);
  else if (ksw % 6 == 0) TG_SK(6);
  else if (ksw % 5 == 0) TG_SK(5);
  else if (ksw % 3 == 0) TG_SK(3);
  else if (ksw % 2 == 0) TG_SK(2);
  else TG_SK(1);
#undef TG_SK
  TG_LAUNCH_CHECK();
  return TG_OK;
}

// single-chunk slices (what the LayerNorm fold needs): k-steps per wave in the instantiated set
inline bool skinny_single_chunk(int ksw) { return ksw == 20 || ksw == 16 || ksw == 12 || ksw == 10 || ksw == 8 || ksw == 6 || ksw == 5 || ksw == 3 || ksw == 2 || ksw == 1; }

template <typename T>
int launch_skinny(const SkinnyParams& p, hipStream_t st) {
  if (p.K % 128 == 0 && (!p.ln || skinny_single_chunk(p.K / 128))) return launch_skinny_nw<T, 8>(p, st);
  return launch_skinny_nw<T, 4>(p, st);
}

}  // namespace

extern "C" int tg_skinny_gemm(const tg_skinny_desc* d, void* stream) {
  TG_CHECK(d != nullptr, TG_ERR_ARG, "tg_skinny_gemm: null descriptor");
  TG_CHECK(d->dtype == TG_BF16 || d->dtype == TG_F16, TG_ERR_ARG, "tg_skinny_gemm: dtype %d", d->dtype);
  TG_CHECK(d->x && d->wpk && d->M > 0, TG_ERR_ARG, "tg_skinny_gemm: null operand or M <= 0");
  TG_CHECK(d->N > 0 && d->N % 32 == 0, TG_ERR_ARG, "tg_skinny_gemm: N = %d must be a positive multiple of 32 (one workgroup per 32-column tile)", d->N);
  TG_CHECK(d->K > 0 && d->K % 64 == 0, TG_ERR_ARG, "tg_skinny_gemm: K = %d must be a positive multiple of 64 (four waves x 16-wide k-steps)", d->K);
  TG_CHECK(d->ldx >= d->K && d->ldx % 8 == 0, TG_ERR_ARG, "tg_skinny_gemm: x row pitch %lld (>= K, multiple of 8 elements)", (long long)d->ldx);
  TG_CHECK(d->act >= TG_ACT_NONE && d->act <= TG_ACT_QUICK_GELU, TG_ERR_ARG, "tg_skinny_gemm: act %d", d->act);
  TG_CHECK(!d->res || d->ldres >= d->N, TG_ERR_ARG, "tg_skinny_gemm: residual pitch");
  const int ksw = d->K / 64;
  if (d->ln) {
    TG_CHECK(d->ln_u && d->ln_v && d->ln_eps > 0.f, TG_ERR_ARG, "tg_skinny_gemm: the LayerNorm fold needs ln_u, ln_v and eps > 0");
    TG_CHECK((d->K % 128 == 0 && skinny_single_chunk(d->K / 128)) || skinny_single_chunk(ksw), TG_ERR_ARG,
             "tg_skinny_gemm: the LayerNorm fold keeps the whole row in registers: K / 64 or K / 128 in {1, 2, 3, 5, 6, 8, 10, 12, 16, 20}, got K = %d", d->K);
  }
  TG_CHECK(d->nseg >= 1 && d->nseg <= 3 && d->rows_per_batch > 0, TG_ERR_ARG, "tg_skinny_gemm: nseg = %d (1..3), rows_per_batch = %d (> 0)", d->nseg, d->rows_per_batch);
  SkinnyParams p;
  p.x = d->x; p.ldx = d->ldx; p.wpk = d->wpk; p.u = d->ln_u; p.v = d->ln_v; p.bias = d->bias; p.res = d->res; p.ldres = d->ldres;
  int prev = 0;
  for (int i = 0; i < 3; ++i) {
    p.seg[i].ptr = nullptr; p.seg[i].ld = 0; p.seg[i].bs = 0; p.seg[i].n_end = 0; p.seg[i].transposed = 0;
    if (i >= d->nseg) continue;
    TG_CHECK(d->seg[i].ptr != nullptr && d->seg[i].n_end > prev && d->seg[i].n_end % 32 == 0, TG_ERR_ARG,
             "tg_skinny_gemm: segment %d: null pointer or n_end = %d not an increasing multiple of 32", i, d->seg[i].n_end);
    p.seg[i].ptr = d->seg[i].ptr; p.seg[i].ld = d->seg[i].ld; p.seg[i].bs = d->seg[i].batch_stride; p.seg[i].n_end = d->seg[i].n_end;
    p.seg[i].transposed = d->seg[i].transposed;
    prev = d->seg[i].n_end;
  }
  TG_CHECK(prev == d->N, TG_ERR_ARG, "tg_skinny_gemm: the segments cover %d of N = %d columns", prev, d->N);
  p.nseg = d->nseg; p.M = (int)d->M; p.N = d->N; p.K = d->K; p.rows_per_batch = d->rows_per_batch; p.act = d->act; p.ln = d->ln ? 1 : 0; p.ln_eps = d->ln_eps;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return d->dtype == TG_BF16 ? launch_skinny<bf16_t>(p, st) : launch_skinny<f16_t>(p, st);
}

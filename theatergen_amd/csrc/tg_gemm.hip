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
#include "tg_gemm_common.h"
#include "tg_gemm_glds.h"

namespace {

struct TileCfg { int bm, bn, bk; };
const TileCfg kTiles[] = {{128, 128, 64}, {64, 64, 64}, {128, 64, 64}, {64, 128, 64}, {128, 128, 64}, {256, 256, 64}, {128, 128, 32},
                          {128, 160, 64}, {128, 160, 64}, {128, 160, 64}};
constexpr int kNumTiles = 10;
// ids 7 / 8 / 9 (round 5, force_tile 21 / 22 / 23): 128 x 160 tiles, four waves of 32 tokens x 160 channels (1 x 5 MFMA tiles, fragments per k-step).
// TILE COUNT, not tile shape, is what they are for: the UNet's mid-level projections are M x N = 16384 x 640 and 4096 x 1280 — 640 / 320 tiles of
// 128 x 128 on 512 (768) co-resident slots = one round at 62 .. 83 % with the busiest CUs holding three tiles, but 512 / 256 tiles of 128 x 160:
// exactly two / one per CU.  7: BK = 64, three stages (108 KB, one workgroup per CU, two K-tiles in flight); 9 (= 8): BK = 64, two stages (72 KB, two per
// CU).  (A BK = 32 / four-stage variant does not exist: 160 weight rows are not a whole number of 16-row DMA instructions per wave.)  Isolated, rotating
// operands, us (profiles/r5_t160_sweep.txt; 128 x 128 -> 128 x 160): 16384 x 640 x 640 + res 30.6 -> 27.5, x 2560 86.4 -> 70.2, x 1280 44.0 -> 35.2;
// 4096 x 1280 x 1280 + res 27.3 -> 24.9, x 5120 87.5 -> 79.2; 65536 x 320 x 1280 97.0 -> 84.0 (N = 320 is 2.5 tiles of 128: a sixth of those MFMAs is padding).
// id 4 = 128x128 with 3 stages (forced only); id 6 = 128x128 with three 32-wide K stages (48 KB: three
// workgroups per CU instead of two; forced / dev switch: see make_plan)
// id 5 = 256x256, 8 waves of 128x64, fragments read per k-step (230 VGPRs): +11..22 % over 128x128 on large plain GEMMs
// (8192x4096x4096 929 vs 839 TF, 16384x5120x2560 1001 vs 818) but no gain at the SD-1.5 UNet's K = 320..1280 with the GEGLU
// epilogue (scripts/dev_big_tile.py), so it is forced-only for now
// force_tile: 1 + tile id (0 = heuristic)

// tile: kTiles id; the first `full` tiles are computed whole, each of the `tail` last tiles is cut into `s` K-ranges of
// `kps` units (K-tiles, or 64-channel chunks for the halo kernel); grid = full + tail * s work items.
struct Plan { int tile; bool halo; int full, tail, s, kps; long tiles_m, tiles_n; };



template <typename T>
__global__ void splitk_reduce_kernel(GemmParams p);

inline int64_t plan_workspace_bytes(const Plan& pl) {
  return pl.s > 1 ? (int64_t)pl.tail * pl.s * kTiles[pl.tile].bm * kTiles[pl.tile].bn * 4 : 0;
}

template <typename T>
int launch_reduce(const GemmParams& p, const Plan& pl, hipStream_t st) {
  if (pl.s > 1) {
    hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned)pl.tail * 8), dim3(256), 0, st, p);
    TG_LAUNCH_CHECK();
  }
  return TG_OK;
}

inline bool halo_eligible(const tg_gemm_desc* d) {
  if (d->mode != 1 || d->stride != 1 || d->pad_mode != 0 || d->force_tile != 0 || d->force_split_k > 1 || d->act != TG_ACT_NONE || d->geglu) return false;
  if (d->c0 % BK != 0 || (d->a1 && d->c1 % BK != 0) || d->M % 128 != 0) return false;
  if (d->out_w == 8) return d->out_h == 8 && !d->upsample && d->M >= 1024;   // two whole 8x8 images per block
  if (d->out_w != 16 && d->out_w != 32 && d->out_w != 64) return false;
  const int th = 128 / d->out_w;
  if (d->out_h % th != 0) return false;
  return d->M >= 4096;      // small-M layers are weight-streaming bound: the K-split tail handles them
}

template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p) {
  // 8 blocks per tail tile (row slices): sum its tail_s fp32 partials in split order, then the regular epilogue
  const int t = blockIdx.x >> 3, slice = blockIdx.x & 7;
  const int lbid = p.full_tiles + t;
  const long m0 = (long)(lbid / p.tiles_n) * p.tile_bm, n0 = (long)(lbid % p.tiles_n) * p.tile_bn;
  const int q4 = p.tile_bn / 4;
  const int rows = p.tile_bm / 8;
  const long tile_elems = (long)p.tile_bm * p.tile_bn;
  const float* base = p.ws + (long)t * p.tail_s * tile_elems;
  for (int q = threadIdx.x; q < rows * q4; q += blockDim.x) {
    const int lr = slice * rows + q / q4, lc = (q % q4) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < p.tail_s; ++z) s += *reinterpret_cast<const f32x4*>(base + z * tile_elems + (long)lr * p.tile_bn + lc);
    // slab conv patch tiles: partial rows are in tile-local order, the token comes from the patch geometry
    const long m = p.patch_pwl > 0 ? patch_token(p, lbid / p.tiles_n, lr) : m0 + lr;
    epilogue_store4<T>(p, m, n0 + lc, s[0], s[1], s[2], s[3]);
  }
}


inline bool halo_eligible(const tg_gemm_desc* d);

// 128x128 tile counts up to this take the 64x64 tile (4x the blocks, three per CU).  128 was fitted on the SD-1.5 bench (CFG batch 16);
// dev knob TG_T64_MAX for the batch-2 plans (BASELINE configs[3] / [4]: 2048 x 1280 x 1280 is 160 tiles = a third of the chip)
inline long t64_max_tiles() {
  const char* e = getenv("TG_T64_MAX");
  return e ? strtol(e, nullptr, 0) : 128;
}

// dev A/B knob TG_T160 (bit mask, default 7): 1 = 128 x 160 tiles for plain GEMMs where they fill whole rounds, 2 = the same for the LayerNorm-folded
// projections, 4 = plain GEMMs whose N is 2.5 / 7.5 tiles of 128 (N = 320, 960)
inline int t160_mode() {
  const char* e = getenv("TG_T160");
  return e ? (int)strtol(e, nullptr, 0) : 7;
}

inline long t3_max_tiles() {
  const char* e = getenv("TG_T3_MAX");
  return e ? strtol(e, nullptr, 0) : 256;
}

Plan make_plan(const tg_gemm_desc* d) {
  // Tile: measured on MI355X over the UNet's shapes (scripts/dev_gemm_bench.py) the 128x128 tile with 2 blocks per CU is
  // the best or within a few % of the best everywhere; skinny problems (one dimension <= 64) take the matching tile.
  const long M = d->M, N = d->N, K = d->K;
  const bool halo = halo_eligible(d);
  int t = 0;
  if (!halo) {
    if (N <= 64 && M > 64) t = 2;        // 128 x 64
    else if (M <= 64 && N > 64) t = 3;   // 64 x 128
    else if (M <= 64 && N <= 64) t = 1;  // 64 x 64
    // few 128x128 tiles (the 8x8 level, M = 1024): 64x64 tiles put 4x as many blocks on the chip (3 per CU):
    // 1024x1280x1280 22 -> 13 us, K = 5120 69 -> 39 us (scripts/dev_tile_sweep.py)
    else if (d->mode == 0 && !d->geglu && ((M + 127) / 128) * ((N + 127) / 128) <= t64_max_tiles()) t = 1;
    // round 3 (batch-2 plans): up to 256 128x128-tiles (under one tile per CU) the 128 x 64 tile: twice the blocks, 3 stages.  Isolated
    // (scripts/dev_tile_sweep_b2.py): 2048 x 1280 x 5120 63 -> 52 us, 2048 x 1280 x 1280 21 -> 18 us, 4608 x 640 x 640 14 -> 11 us; in the
    // graph-replayed steps (same-box A/B, TG_T3_MAX 0 / 256): configs[4] 27.76 -> 26.73 ms/step, configs[3] 11.49 -> 11.39 ms/step.  No SD-1.5
    // CFG-batch-16 shape falls in the range (its 16 x 16 level is 320 tiles).
    else if (d->mode == 0 && !d->geglu && N <= 1280 && ((M + 127) / 128) * ((N + 127) / 128) <= t3_max_tiles()) t = 2;
    // dev A/B (TG_GEMM_FLAGS bit 14): short-K plain GEMMs on the 32-wide K stages (three co-resident workgroups per CU)
    // short-K plain GEMMs (K <= 640: the 64x64 / 32x32 levels' attention and proj_in / proj_out projections) take the 128x128 tile
    // on three 32-wide K stages: 48 KB of LDS = THREE co-resident workgroups per CU instead of two, more prologue / epilogue
    // latency of one block under another's K loop.  Same K order: bit-identical results.  Graph-replay A/B, three interleaved rounds
    // (scripts/dev_env_ab.sh TG_T7_MAXK "0 640 1280 5120"): 8.328 -> 8.350 images/s at 640 (+0.27 %), 8.302 at 1280, 8.299 at 5120.
    {
      const char* mk = getenv("TG_T7_MAXK");           // dev knob
      const long maxk = mk ? strtol(mk, nullptr, 0) : 640;
      if (t == 0 && d->mode == 0 && !d->geglu && K <= maxk && K % 32 == 0) t = 6;
      // round 3 (dev switch TG_T7_FIT=1, A/B in profiles/r3_gemm_findings.md): longer-K plain GEMMs whose 128x128 tile count fits ONE
      // round of the three-workgroup variant (768 slots) but not one round of the two-workgroup one (512): 16384 x 640 x 2560
      // (FeedForward net.2 of the 32 x 32 level) is 640 tiles = a full round + a quarter-filled one on 512 slots
      {
        const long t128 = ((M + 127) / 128) * ((N + 127) / 128);
        const char* fit = getenv("TG_T7_FIT");
        if (fit && fit[0] == '1' && t == 0 && d->mode == 0 && !d->geglu && K % 32 == 0 && t128 > 512 && t128 <= 768) t = 6;
      }
    }
    // round 5: tile-count-aware 128 x 160 tiles (see kTiles) — where they fill whole rounds and the 128 x 128 tiling does not
    if ((t == 0 || t == 6) && d->mode == 0 && !d->geglu && d->act == TG_ACT_NONE && d->n_split <= 0 && N % 160 == 0 && M % 128 == 0 && K % 64 == 0 && (t160_mode() & 5)) {
      const long t128 = ((M + 127) / 128) * ((N + 127) / 128), s128 = (t == 6) ? 768 : 512;
      const long t160 = (M / 128) * (N / 160);
      const double eff128 = (double)t128 / (double)(((t128 + s128 - 1) / s128) * s128);
      const double e256 = (double)t160 / (double)(((t160 + 255) / 256) * 256), e512 = (double)t160 / (double)(((t160 + 511) / 512) * 512);
      const bool one_per_cu = e256 > e512 + 1e-9;              // whole rounds only at one workgroup per CU (three K stages): 256 / 768 / 1280 tiles
      const double eff160 = one_per_cu ? e256 : e512;
      const bool ragged128 = N % 128 != 0 && K >= 640 && (t160_mode() & 4);          // N = 320 / 960: the last 128-column tile is half padding
      if (d->force_split_k <= 1 && t160 >= 192 && (((t160_mode() & 1) && eff160 >= eff128 + 0.1) || (ragged128 && eff160 >= eff128 - 0.01))) t = one_per_cu ? 7 : 9;
    }
    if (d->force_tile >= 21 && d->force_tile <= 23) t = d->force_tile - 14;
    else if (d->force_tile > 0) t = d->force_tile - 1;
    if (t >= kNumTiles || t < 0) t = 0;
  }
  const long tm = (M + kTiles[t].bm - 1) / kTiles[t].bm, tn = (N + kTiles[t].bn - 1) / kTiles[t].bn;
  const long T = tm * tn;
  // K units that a split may cut at, and the fewest a work item should keep
  const int units = halo ? (int)((d->c0 + (d->a1 ? d->c1 : 0)) / BK) : (int)((K + kTiles[t].bk - 1) / kTiles[t].bk);
  const int min_units = halo ? 2 : 8;
  // Tail split.  The grid runs in rounds of S co-resident blocks (LDS-limited: 2 per CU for the 64 KB tiles).  A last
  // round that fills only part of the chip (640 tiles on 512 slots: the 32x32 layers; 320 or 80 tiles: 16x16 / 8x8) can
  // be cut along K so that its work spreads over every CU:
  //     cost(c) = rounds(c) * (units / c * t_unit + t_fix) + t_reduce(c),   t_reduce = 6 us + 0.065 us per partial tile
  // against the unsplit tail, which runs ~0.6x as long as a full round when at most one block per CU is left (no
  // co-resident block to share the matrix pipe / L2 path with).  Constants from scripts/dev_gemm_ksweep.py and
  // scripts/dev_split_ab.py on MI355X (us; only ratios matter).  In practice this splits the 8x8 weight-streaming
  // convs (~6 ways: 204 -> 91 us), the K >= 11520 halo convs of the 32x32 / 16x16 levels (-16 .. -20 %) and the
  // longest-K 8x8 projections; every other layer measured faster unsplit (partials cost more than the idle CUs).
  long S = 512;
  if (!halo && (t == 1 || t == 6)) S = 768;
  if (!halo && (t == 4 || t == 5 || t == 7)) S = 256;
  long full = (T / S) * S, rem = T - full;
  int s = 1;
  if (d->force_split_k > 0) {
    full = 0; rem = T; s = d->force_split_k;
  } else if (rem > 0 && !d->geglu && units >= 2 * min_units) {
    const double t_unit = halo ? 9.0 : (d->mode == 1 ? 1.5 : 1.1), t_fix = 6.0;
    // an unsplit tail that leaves at most one block per CU runs faster than a full round — much faster for the GEMM and
    // halo kernels (0.6x), hardly for the implicit-GEMM conv whose blocks are bound by their own DMA latency (0.85x)
    const double unsplit = (units * t_unit + t_fix) * (2 * rem <= S ? ((!halo && d->mode == 1) ? 0.85 : 0.6) : 1.0);
    double best = 1e30;
    int best_c = 1;
    for (int c = 2; c <= 8 && units / c >= min_units; ++c) {
      const int kps = (units + c - 1) / c;
      const long rounds = (rem * c + S - 1) / S;
      const double cost = rounds * (kps * t_unit + t_fix) + 6.0 + 0.065 * (double)(rem * c);
      if (cost < best) { best = cost; best_c = c; }
    }
    if (best < unsplit * 0.95) s = best_c;          // a split must clearly pay for its partial traffic
  }
  if (s > units) s = units;
  if (s < 1) s = 1;
  int kps = (units + s - 1) / s;
  s = (units + kps - 1) / kps;
  if (s <= 1) { s = 1; full = T; rem = 0; kps = units; }
  return Plan{t, halo, (int)full, (int)rem, s, kps, tm, tn};
}

template <typename T, int BM, int BN, int WM, int WN, bool CONV, int STAGES, int BKT, int EPI>
void launch_glds(const GemmParams& p, dim3 grid, size_t lds, hipStream_t st) {
  auto k = gemm_glds_kernel<T, BM, BN, WM, WN, CONV, STAGES, BKT, EPI>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr;
  hipLaunchKernelGGL(k, grid, dim3(WM * WN * 64), lds, st, p);
}

template <typename T, int BM, int BN, int WM, int WN, int STAGES, int BKT = 64>
int launch_cfg2(const tg_gemm_desc* d, const GemmParams& p, const Plan& pl, hipStream_t st) {
  const size_t lds = (size_t)STAGES * (BM + BN) * BKT * sizeof(T);
  dim3 grid((unsigned)(pl.full + pl.tail * pl.s));
  // epilogue kind: 0 = linear only, 1 = generic (activation / GEGLU on any tile), 2 = GEGLU on the default plain tile
  constexpr bool kMainTile = (BM == 128 && BN == 128 && STAGES == 2) || (BM == 256 && BN == 256);   // tiles with a GEGLU-only instance
  const int epi = d->geglu ? ((kMainTile && d->mode != 1) ? 2 : 1) : (d->act == TG_ACT_NONE ? 0 : 1);
  if (d->mode == 1) {
    if (epi == 0) launch_glds<T, BM, BN, WM, WN, true, STAGES, BKT, 0>(p, grid, lds, st);
    else launch_glds<T, BM, BN, WM, WN, true, STAGES, BKT, 1>(p, grid, lds, st);
  } else {
    if (epi == 0) launch_glds<T, BM, BN, WM, WN, false, STAGES, BKT, 0>(p, grid, lds, st);
    else if (epi == 2) {
      if constexpr (kMainTile) launch_glds<T, BM, BN, WM, WN, false, STAGES, BKT, 2>(p, grid, lds, st);
    } else launch_glds<T, BM, BN, WM, WN, false, STAGES, BKT, 1>(p, grid, lds, st);
  }
  TG_LAUNCH_CHECK();
  return launch_reduce<T>(p, pl, st);
}

}  // namespace
// big-tile kernel (tg_gemm_bt.hip): bt_tile 2 = 256 x 256 (the 128 x 320 instance, bt_tile 1, was removed in round 5)
int tg_gemm_bt_launch(const tg_gemm_desc* d, const void* params, int bt_tile, void* stream);
// slab conv kernel (tg_conv_slab.hip): BM x 320 output tiles, GroupNorm(+SiLU) prologue on the staged window
int tg_conv_slab_launch(const tg_gemm_desc* d, const void* params, int splits, void* stream);
bool tg_conv_slab_is_pp(const tg_gemm_desc* d, int patch_pwl, int patch_np, int epi_lds);
// LayerNorm-fused projections (tg_gemm_ln.hip): 128 x 128 tiles, no K split
int tg_gemm_ln_launch(const tg_gemm_desc* d, const void* params, int short_k, int grid, void* stream);
// 128 x 160 tiles (tg_gemm_t160.hip): variant 0 = BK 64 x 3 stages, 1 = BK 32 x 4 stages, 2 = BK 64 x 2 stages
int tg_gemm_t160_launch(const tg_gemm_desc* d, const void* params, int variant, int grid, void* stream);
// LDS-halo conv (tg_conv_halo.hip): grid = full tiles + tail tiles * K splits
int tg_conv_halo_launch(const tg_gemm_desc* d, const void* params, int grid, void* stream);
// ping-pong 256 x 256 tiles (tg_gemm_pp.hip, round 6): 8 waves in two groups one barrier apart, persistent
int tg_gemm_pp_launch(const tg_gemm_desc* d, const void* params, void* stream);
// the same structure on 256 x 160 tiles (tg_gemm_pp160.hip): the N = 640 / 1920 / 320 projections
int tg_gemm_pp160_launch(const tg_gemm_desc* d, const void* params, void* stream);
namespace {

// Tile geometry of the slab kernel for an out_h x out_w map: patch width *pw and patches per 128-pixel tile *np (tg_conv_slab.hip).
// Whole image rows for the 64 / 32 / 16-wide maps (*patch = false); wider or odd maps are cut into patches: multiples of 64 -> 2 x 64,
// of 32 -> 4 x 32 (SD-2.1's 96), of 16 -> 8 x 16 (48, 80), of 8 -> two 8 x 8 patches per tile (the 8 x 8 level, 24, 40).
inline bool slab_geometry(const tg_gemm_desc* d, int* pw, int* np, bool* patch) {
  const int w = d->out_w, h = d->out_h;
  int P = 0, NPv = 1;
  if (w == 64 || w == 32 || w == 16) P = w;
  else if (w % 64 == 0) P = 64;
  else if (w % 32 == 0) P = 32;
  else if (w % 16 == 0) P = 16;
  else if (w % 8 == 0) { P = 8; NPv = 2; }
  else return false;
  const int th = 128 / (P * NPv);
  if (h % th != 0 || d->M % 128 != 0) return false;
  *pw = P; *np = NPv; *patch = (P != w) || NPv > 1;
  return true;
}

// Slab conv (tg_conv_slab.hip): stride-1 pad-1 convs with N a multiple of 320 on 16 / 32 / 64-wide maps, 128-pixel x 320-channel
// tiles.  -> K splits per tile (over 64-channel chunks), 0 = not taken.  force_tile 11 / 12 = 1 / 2 splits regardless of the tile
// count (tests); the heuristic wants the persistent grid (one workgroup per CU, 256) at least 3/4 full in every round, splitting
// the channel chunks in 2 if that is what it takes (the 16-wide maps: 128 tiles; every split keeps >= 5 chunks = 45 K-steps).
// TG_GEMM_FLAGS bit 7 (dev) turns it off.  (A 64 x 320 tile for the 16-wide maps was built, measured and dropped: 40 KB of weights
// per 640 matrix-pipe cycles = 64 B/clk per CU is the L2's whole bandwidth: 52 ms against the halo kernel's 36 per 21 UNet calls.)
inline int slab_splits_of(const tg_gemm_desc* d) {
  if (d->mode != 1 || d->stride != 1 || d->upsample || d->pad_mode != 0 || d->act != TG_ACT_NONE || d->geglu) return 0;
  if (d->N % 320 != 0 || d->c0 % BK != 0 || (d->a1 && d->c1 % BK != 0) || d->n_split > 0) return 0;
  int pw, np;
  bool patch;
  if (!slab_geometry(d, &pw, &np, &patch)) return 0;
  const int chunks = (d->c0 + (d->a1 ? d->c1 : 0)) / BK;
  // force_tile 11: the slab kernel with force_split_k (default 1) splits, 12: two splits (tests / dev sweeps)
  if (d->force_tile == 11) {
    if (d->force_split_k <= 1) return 1;
    if (d->force_split_k > chunks) return 0;
    const int cps = (chunks + d->force_split_k - 1) / d->force_split_k;
    return (chunks + cps - 1) / cps;                                // every split keeps at least one chunk
  }
  if (d->force_split_k > 1) return 0;
  if (d->force_tile == 12) return chunks >= 2 ? 2 : 0;
  if (d->force_tile != 0) return 0;
  long flags = 0;
  { const char* e = getenv("TG_GEMM_FLAGS"); flags = e ? strtol(e, nullptr, 0) : 0; }
  const bool old_patch = (d->out_w == 128 || d->out_w == 96);     // round-3 first pass: 2 x 64 / 4 x 32 patches, never split
  if ((flags & 128) || (old_patch && (flags & 1024))) return 0;    // dev A/B: bit 7 no slab kernel, bit 10 no patch tiles
  const long t = (d->M / 128) * (d->N / 320);
  auto full = [](long n) { return 4 * n >= 3 * ((n + 255) / 256) * 256; };
  if (!patch || old_patch) {
    // a patch-tile layer has no 128-pixel halo kernel to fall back to (the implicit-GEMM conv re-fetches every window 9 times): half a
    // chip of slab tiles already beats it (BASELINE configs[3]: 96 x 96 x batch 2 = 144 tiles)
    if (old_patch && t >= 128) return 1;
    if (!old_patch && full(t)) return 1;
    if (!old_patch && chunks >= 10 && full(2 * t)) return 2;
  }
  // Round 3, second pass: what the rules above used to leave to the IMPLICIT-GEMM conv (which re-fetches every window 9 times) — the
  // small-M levels of the batch-2 plans: SD-2.1 48 x 48 (3 x 16 patches) and 24 x 24 (two 8 x 8 patches per tile), SDXL 32 x 32 at
  // batch 2, SD-1.5 32 x 32 / 16 x 16 at batch 2 — with a K split over the channel chunks that puts enough work items on the chip.
  // Layers the LDS-halo kernel takes (power-of-two widths with M >= 4096, the 8 x 8 level at M >= 1024) stay there: measured
  // (scripts/dev_slab_split_sweep.py, profiles/r3_slab_split_sweep.txt) 51 vs 55 us on SDXL's 64 x 64 320 -> 640 and 57 vs 62 us on
  // the 8 x 8 level at CFG batch 16.  Cost of S splits, one work item per CU and round (us; fitted on the same sweep, only ratios matter):
  //   rounds(t S / 256) x 9 x chunks-per-split x t_k + [S > 1] (t_red + t_part x t S),  t_k = max(0.9, min(t S, 256) / 210)
  // — a K-step takes 0.8 .. 0.9 us on a part-filled chip and the whole chip completes ~210 K-steps per us (72 / 144 / 216 / 252 work
  // items: 0.80 / 0.90 / 1.03 / 1.27 us), so filling the last 20 % of the CUs buys nothing; the fp32 partials are nearly free.
  if (flags & 2048) return 0;                                      // dev A/B: bit 11 = the rules above only
  if (halo_eligible(d) && !(flags & 4096)) return 0;               // dev A/B: bit 12 = this rule for halo-eligible layers too
  const double t_red = 8.0, t_part = 0.02;
  double best = 1e30;
  int best_s = 0;
  for (int c = 1; c <= 8; ++c) {
    const int cps = (chunks + c - 1) / c;
    if (c > 1 && (cps < 2 || (chunks + cps - 1) / cps != c)) continue;
    const long items = t * c;
    const long rounds = (items + 255) / 256;
    const double per_round = (double)(items < 256 ? items : 256);
    const double t_k = per_round / 210.0 > 0.9 ? per_round / 210.0 : 0.9;
    const double cost = rounds * 9.0 * cps * t_k + (c > 1 ? t_red + t_part * (double)items : 0.0);
    if (cost < best) { best = cost; best_s = c; }
  }
  if (t * best_s < 96) return 0;                                   // under ~a third of the chip even when split: not this kernel's case
  return best_s;
}

// (Round 2's loader / compute GEMM on 128 x 320 tiles, tg_gemm_lc.hip / force_tile 13, 14 — faster than the 128 x 128 kernel in isolation on the long-K
// FeedForward projections (16384 x 640 x 2560: 81 vs 102 us), never selected: no gain under graph replay — was REMOVED in round 5: the tile-count-aware
// 128 x 160 tiles run the same shape in 69 us and ARE selected.)
// Big tile (tg_gemm_bt.hip): force_tile 10 = 256 x 256 (round 2 also had force_tile 9 = 128 x 320; the notes below are its measurements); plain GEMM with one A source, K a multiple of 64,
// no K split.  What the heuristic (force_tile 0) takes, and why so little (profiles/r2_gemm_findings.md, all on MI355X):
//   * isolated launches (scripts/dev_bt_bench.py, rotating operands; us, 128x128 -> big tile): fused GEGLU 65536x2560x320
//     230 -> 188, 16384x5120x640 193 -> 156; projections 16384x640x640 29.1 -> 25.6, 16384x640x2560 92 -> 80, 65536x320x320
//     31 -> 29; big GEMMs 8192x4096x4096 723 -> 1016 TF.  In the eager UNet step (scripts/dev_insitu_gemm.py) the same launches
//     save 0.36 ms of 15.1 ms.
//   * in the hipGraph-replayed bench the picture turns: with the 128 x 320 projections ON the whole bench is 2.1 % SLOWER
//     (7.516 vs 7.674 images/s, three interleaved rounds), with only the GEGLU launches on the 256 x 256 tile it is 0.3 %
//     faster (7.698).  rocprofv3 + rocm-smi of the two runs: the projection launches take the same time as before (32.4 us
//     average against 128x128's mix), but the shader clock settles at ~2150 MHz instead of ~2225 MHz (at LOWER package power,
//     1170 vs 1240 W) and every other kernel of the step slows down with it (attention 288 -> 303 us, halo convs +2..4 %).
//   So only the GEGLU tile is selected; 128 x 320 was removed in round 5 (the 128 x 160 tiles took its place AND pay under graph replay).
inline int bt_tile_of(const tg_gemm_desc* d) {
  const int ft = d->force_tile;
  const bool can = d->mode == 0 && d->force_split_k <= 1 && d->a1 == nullptr && d->K % BK == 0;
  if (ft == 10) return can ? 2 : -1;
  if (ft == 9) return -1;                     // (the 128 x 320 big tile of round 2 was removed in round 5: never selected)
  if (ft != 0 || !can) return -1;
  int devf = 0;
  { const char* e = getenv("TG_GEMM_FLAGS"); devf = e ? (int)strtol(e, nullptr, 0) : 0; }   // dev A/B switches
  if (devf & 8) return -1;
  if (d->geglu) {
    const long tiles = ((d->M + 255) / 256) * ((d->N + 255) / 256);
    return (d->M >= 16384 && tiles >= 1024) ? 2 : -1;
  }
  return -1;
}

// Ping-pong 256 x 256 tiles (tg_gemm_pp.hip; force_tile 24): plain single-source GEMMs with M, N multiples of 256 and K of 64 whose tile count fills
// the persistent grid's rounds; linear / activation / GEGLU epilogues, V^T columns on a 64-column boundary, the LayerNorm fold only with precomputed
// row statistics (ln_rows).  Dev A/B knob TG_PP (bit mask, default 15): 1 = GEGLU launches, 2 = linear / activation launches, 4 = LayerNorm-folded ones, 8 = the 256 x 160 tiles (force_tile 25).
inline int pp_mode() {
  const char* e = getenv("TG_PP");
  return e ? (int)strtol(e, nullptr, 0) : 15;
}
inline bool pp_eligible(const tg_gemm_desc* d) {
  if (d->mode != 0 || d->a1 != nullptr || d->force_split_k > 1 || d->a_coef != nullptr) return false;
  if (d->M % 256 != 0 || d->N % 256 != 0 || d->K % 64 != 0 || d->K < 128) return false;
  if (d->n_split > 0 && (d->n_split % 64 != 0 || d->rows_per_batch <= 0 || (d->M / d->rows_per_batch) * (d->N - d->n_split) * d->ldt >= (1LL << 31))) return false;
  if (d->ln_u != nullptr && d->ln_rows == nullptr) return false;
  if (d->geglu && d->n_split > 0) return false;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const long lda = d->lda > 0 ? d->lda : d->c0, ldw = d->ldw > 0 ? d->ldw : d->K;
  if (!(al16(d->a0) && al16(d->w) && al16(d->out) && al16(d->bias) && al16(d->bvec) && al16(d->res)) || lda % 8 != 0 || ldw % 8 != 0 || d->ldc % 8 != 0) return false;
  if ((d->bvec != nullptr && d->ldbvec % 8 != 0) || (d->res != nullptr && d->ldres % 8 != 0)) return false;
  if (d->a_rows_per_batch > 0 && d->a_batch_stride % 8 != 0) return false;
  return true;
}
inline bool pp_selected(const tg_gemm_desc* d) {
  if (d->force_tile == 24) return pp_eligible(d);
  if (d->force_tile != 0 || !pp_eligible(d)) return false;
  const int mode = pp_mode();
  if (d->ln_u != nullptr ? !(mode & 4) : (d->geglu ? !(mode & 1) : !(mode & 2))) return false;
  const long tiles = (d->M / 256) * (d->N / 256);
  const double eff = (double)tiles / (double)(((tiles + 255) / 256) * 256);
  // measured (scripts/dev_gemm8.py, profiles/r6_pp_gemm.txt): the ping-pong loop wins where a workgroup's K loop is long enough to pay for its
  // prologue and the grid fills its rounds (2048 x 10240 x 1280 GEGLU, 320 tiles = 1.25 rounds: 81.8 -> 64.6 us; 8192 x 5120 x 640, 640 tiles: 89.6 -> 67.8);
  // short-K / ragged-N projections stay on the 128 x 160 / 128 x 128 tiles
  return tiles >= 192 && eff >= 0.6 && d->K >= 640;
}

// Ping-pong 256 x 160 tiles (tg_gemm_pp160.hip; force_tile 25): the same problems with N a multiple of 160 instead of 256 (no GEGLU; V^T columns on an
// 80-column boundary) whose 256 x 160 tiles fill whole rounds of the chip: 16384 x 640 (256 tiles), 16384 x 1920 (768), 65536 x 320 (512).
inline bool pp160_eligible(const tg_gemm_desc* d) {
  if (d->mode != 0 || d->a1 != nullptr || d->force_split_k > 1 || d->a_coef != nullptr || d->geglu) return false;
  if (d->M % 256 != 0 || d->N % 160 != 0 || d->K % 64 != 0 || d->K < 128) return false;
  if (d->n_split > 0 && (d->n_split % 80 != 0 || d->rows_per_batch <= 0 || (d->M / d->rows_per_batch) * (d->N - d->n_split) * d->ldt >= (1LL << 31))) return false;
  if (d->ln_u != nullptr && d->ln_rows == nullptr) return false;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const long lda = d->lda > 0 ? d->lda : d->c0, ldw = d->ldw > 0 ? d->ldw : d->K;
  if (!(al16(d->a0) && al16(d->w) && al16(d->out) && al16(d->bias) && al16(d->bvec) && al16(d->res)) || lda % 8 != 0 || ldw % 8 != 0 || d->ldc % 8 != 0) return false;
  if ((d->bvec != nullptr && d->ldbvec % 8 != 0) || (d->res != nullptr && d->ldres % 8 != 0)) return false;
  if (d->a_rows_per_batch > 0 && d->a_batch_stride % 8 != 0) return false;
  return true;
}
inline bool pp160_selected(const tg_gemm_desc* d) {
  if (d->force_tile == 25) return pp160_eligible(d);
  if (d->force_tile != 0 || !pp160_eligible(d) || !(pp_mode() & 8)) return false;
  if (d->ln_u != nullptr && !(pp_mode() & 4)) return false;
  const long tiles = (d->M / 256) * (d->N / 160);
  const double eff = (double)tiles / (double)(((tiles + 255) / 256) * 256);
  return tiles >= 192 && eff >= 0.74 && d->K >= 640;          // (K = 320: five K-tiles, not measured: stays on the 128 x 160 / 128 x 128 tiles)
}

// LayerNorm-folded projections on 128 x 160 tiles: 0 = no, 160 = three stages / one workgroup per CU, 161 = two stages / two per CU (tg_gemm_ln.hip)
inline int ln_t160_of(const tg_gemm_desc* d) {
  // (attn2.to_q only: measured in situ, same box — profiles/r5_t160_findings.md — the q | k | v^T projections, whose V^T third leaves through the
  // transposed direct epilogue, are no faster on these tiles: 16384 x 1920 x 640 76.4 -> 80.9 us, 4096 x 3840 x 1280 74.5 -> 90.3 us; to_q 33.0 -> 29.4, 29.3 -> 27.1)
  if (!(t160_mode() & 2) || d->geglu || d->N % 160 != 0 || d->M % 128 != 0 || d->K % 64 != 0 || d->n_split > 0) return 0;
  const long t128 = (d->M / 128) * ((d->N + 127) / 128), s128 = d->K <= 640 ? 768 : 512;
  const long t160 = (d->M / 128) * (d->N / 160);
  const double eff128 = (double)t128 / (double)(((t128 + s128 - 1) / s128) * s128);
  const double e256 = (double)t160 / (double)(((t160 + 255) / 256) * 256), e512 = (double)t160 / (double)(((t160 + 511) / 512) * 512);
  const bool one_per_cu = e256 > e512 + 1e-9;
  const double eff160 = one_per_cu ? e256 : e512;
  if (t160 < 192 || eff160 < eff128 + 0.05) return 0;
  return one_per_cu ? 160 : 161;
}

// operands / strides allow the LDS-transposed, 16-byte-coalesced epilogue (GemmParams::epi_lds)
inline bool epi_lds_of(const tg_gemm_desc* d) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return d->N % 8 == 0 && d->ldc % 8 == 0 && al16(d->out) && al16(d->bias) && al16(d->bvec) && al16(d->res) &&
         (d->bvec == nullptr || d->ldbvec % 8 == 0) && (d->res == nullptr || d->ldres % 8 == 0) && (d->n_split == 0 || d->n_split % 64 == 0);
}

// tg_gemm_desc.out_gn_partials: only the two-wave slab kernel's unsplit epilogue writes them (every compute wave owns 64 pixels x 80 channels: whole groups,
// one batch item).  -> 64-pixel blocks per batch item, 0 = this descriptor's kernel cannot.
inline int gn_partial_blocks_of(const tg_gemm_desc* d) {
  if (d->out_gn_groups <= 0 || d->N % d->out_gn_groups != 0 || 80 % (d->N / d->out_gn_groups) != 0) return 0;
  if (slab_splits_of(d) != 1) return 0;
  int pw = 0, np = 1;
  bool patch = false;
  if (!slab_geometry(d, &pw, &np, &patch) || patch || np != 1) return 0;
  if (!tg_conv_slab_is_pp(d, 0, 1, epi_lds_of(d))) return 0;
  const long hw = (long)d->out_h * d->out_w;
  if (hw % 64 != 0) return 0;
  return (int)(hw / 64);
}

template <typename T>
int launch_gemm(const tg_gemm_desc* d, hipStream_t st) {
  Plan pl = make_plan(d);
  GemmParams p{};
  p.a0 = d->a0; p.a1 = d->a1; p.c0 = d->c0; p.c1 = d->a1 ? d->c1 : 0;
  p.in_h = d->in_h; p.in_w = d->in_w; p.out_h = d->out_h; p.out_w = d->out_w;
  p.stride = d->stride; p.upsample = d->upsample; p.pad_lo = d->pad_mode == 1 ? 0 : 1;
  p.w = d->w; p.M = d->M; p.N = d->N; p.K = d->K;
  p.bias = d->bias; p.bvec = d->bvec; p.ldbvec = d->ldbvec;
  p.rows_per_batch = d->rows_per_batch > 0 ? d->rows_per_batch : d->M;
  p.res = d->res; p.ldres = d->ldres; p.act = d->act; p.geglu = d->geglu; p.out_scale = d->out_scale;
  p.out = d->out; p.ldc = d->ldc; p.n_split = d->n_split; p.out_t = d->out_t; p.ldt = d->ldt;
  p.ws = reinterpret_cast<float*>(d->workspace);
  p.full_tiles = pl.full; p.tail_s = pl.s; p.kt_per_split = pl.kps; p.tiles_n = (int)pl.tiles_n;
  p.tile_bm = kTiles[pl.tile].bm; p.tile_bn = kTiles[pl.tile].bn;
  p.a_rpb = d->mode == 0 ? d->a_rows_per_batch : 0; p.a_bs = d->a_batch_stride;
  p.lda = d->lda > 0 ? d->lda : p.c0; p.ldw = d->ldw > 0 ? d->ldw : d->K;
  {
    const char* e = getenv("TG_GEMM_FLAGS");          // dev experiments; read per launch so one process can A/B
    p.flags = e ? (int)strtol(e, nullptr, 0) : 0;
  }
  p.a_coef = d->a_coef; p.a_silu = d->a_silu;
  p.patch_pwl = 0; p.patch_np = 1;
  p.ln_u = d->ln_u; p.ln_v = d->ln_v; p.ln_eps = d->ln_eps; p.ln_rows = d->ln_rows;
  p.epi_lds = epi_lds_of(d);
  p.gn_part = d->out_gn_partials; p.gn_cpg = d->out_gn_partials ? (int)(d->N / d->out_gn_groups) : 0;
  TG_CHECK(d->out_gn_partials == nullptr || gn_partial_blocks_of(d) > 0, TG_ERR_UNSUPPORTED,
           "tg_gemm: out_gn_partials needs an unsplit stride-1 conv on the two-wave slab kernel with 80 %% (N / groups) == 0 (ask tg_gemm_gn_partial_blocks)");
  if (pp_selected(d)) return tg_gemm_pp_launch(d, &p, st);
  if (pp160_selected(d)) return tg_gemm_pp160_launch(d, &p, st);
  TG_CHECK(d->force_tile != 25, TG_ERR_UNSUPPORTED, "tg_gemm: force_tile 25 (ping-pong 256 x 160 tiles) needs a plain single-source GEMM with M %% 256 == 0, N %% 160 == 0, K %% 64 == 0, 16-byte aligned operands and no GEGLU");
  TG_CHECK(d->force_tile != 24, TG_ERR_UNSUPPORTED, "tg_gemm: force_tile 24 (ping-pong 256 x 256 tiles) needs a plain single-source GEMM with M %% 256 == 0, N %% 256 == 0, K %% 64 == 0 and 16-byte aligned operands");
  if (d->ln_u != nullptr) {
    // LayerNorm-fused projection: whole rows per workgroup (no K split), 128 x 128 tiles — or 128 x 160 where those fill whole rounds — in XCD-chunked order
    const int t160 = ln_t160_of(d);
    const int bn = t160 ? 160 : 128;
    const long tiles = ((d->M + 127) / 128) * ((d->N + bn - 1) / bn);
    p.full_tiles = (int)tiles; p.tail_s = 1; p.tiles_n = (int)((d->N + bn - 1) / bn); p.tile_bm = 128; p.tile_bn = bn;
    p.kt_per_split = 0;
    return tg_gemm_ln_launch(d, &p, t160 ? t160 : (d->K <= 640 ? 1 : 0), (int)tiles, st);
  }
  if (const int sp = slab_splits_of(d); sp > 0) {
    const long tiles = (d->M / 128) * (d->N / 320);
    if (sp > 1) {
      const int64_t need = tiles * sp * 128 * 320 * 4;
      TG_CHECK(d->workspace != nullptr && d->workspace_bytes >= need, TG_ERR_ARG, "tg_gemm conv: the K split needs %lld workspace bytes, got %lld",
               (long long)need, (long long)d->workspace_bytes);
    }
    {
      int pw = 0, np = 1;
      bool patch = false;
      slab_geometry(d, &pw, &np, &patch);
      p.patch_np = np;
      if (patch) { int l = 0; while ((1 << l) < pw) ++l; p.patch_pwl = l; }
    }
    int rc = tg_conv_slab_launch(d, &p, sp, st);
    if (rc != TG_OK || sp == 1) return rc;
    p.tiles_n = (int)(d->N / 320); p.full_tiles = 0; p.tail_s = sp; p.tile_bm = 128; p.tile_bn = 320;
    Plan rp = pl;
    rp.tail = (int)tiles; rp.s = sp;
    return launch_reduce<T>(p, rp, st);
  }
  TG_CHECK(d->a_coef == nullptr, TG_ERR_UNSUPPORTED, "tg_gemm: a_coef (GroupNorm prologue) needs a problem the slab conv kernel takes (tg_gemm_plan kernel_kind 4)");
  {
    const int64_t need = plan_workspace_bytes(pl);
    TG_CHECK(need == 0 || (d->workspace != nullptr && d->workspace_bytes >= need), TG_ERR_ARG,
             "tg_gemm: the K-split tail needs %lld workspace bytes, got %lld", (long long)need, (long long)d->workspace_bytes);
  }
  if (const int bt = bt_tile_of(d); bt >= 0) {
    TG_CHECK(d->lda <= 0 && d->ldw <= 0, TG_ERR_UNSUPPORTED, "tg_gemm: lda / ldw are taken by the 128 x 128 / 128 x 160 plain kernels only");
    TG_CHECK(!d->geglu || bt == 2, TG_ERR_ARG, "tg_gemm: the GEGLU epilogue needs the 256 x 256 big tile (force_tile 10)");
    TG_CHECK(d->n_split <= 0 || d->n_split % 128 == 0, TG_ERR_ARG, "tg_gemm: the big tile needs n_split on a wave-tile boundary");
    return tg_gemm_bt_launch(d, &p, bt, st);
  }
  if (pl.halo) {
    const int rc = tg_conv_halo_launch(d, &p, pl.full + pl.tail * pl.s, st);
    if (rc != TG_OK) return rc;
    return launch_reduce<T>(p, pl, st);
  }
  switch (pl.tile) {
    case 0: return launch_cfg2<T, 128, 128, 2, 2, 2>(d, p, pl, st);
    case 1: return launch_cfg2<T, 64, 64, 2, 2, 3>(d, p, pl, st);
    case 2: return launch_cfg2<T, 128, 64, 4, 1, 3>(d, p, pl, st);
    case 3: return launch_cfg2<T, 64, 128, 1, 4, 3>(d, p, pl, st);
    case 4: return launch_cfg2<T, 128, 128, 2, 2, 3>(d, p, pl, st);   // 3 stages, 96 KB: 1 block / CU (forced only)
    case 6: return launch_cfg2<T, 128, 128, 2, 2, 3, 32>(d, p, pl, st);   // three 16 KB K stages: 3 blocks / CU
    case 7: case 8: case 9: {                                             // 128 x 160 tiles (tg_gemm_t160.hip): 7 = 1 block / CU, 8 / 9 = 2 blocks / CU
      TG_CHECK(d->mode == 0 && !d->geglu && d->act == TG_ACT_NONE, TG_ERR_ARG, "tg_gemm: the 128 x 160 tiles take plain GEMMs with a linear epilogue");
      const int rc = tg_gemm_t160_launch(d, &p, pl.tile - 7, pl.full + pl.tail * pl.s, st);
      if (rc != TG_OK) return rc;
      return launch_reduce<T>(p, pl, st);
    }
    default: return launch_cfg2<T, 256, 256, 2, 4, 2>(d, p, pl, st);   // 8 waves of 128x64, 128 KB, 1 block / CU
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
  if (d->geglu) {
    TG_CHECK(d->N % 64 == 0 && d->n_split <= 0 && !d->bvec && !d->res && d->act == TG_ACT_NONE && d->force_split_k <= 1,
             TG_ERR_ARG, "tg_gemm: GEGLU epilogue needs N %% 64 == 0 (packed a|gate groups) and no other epilogue terms");
    const int ft = d->force_tile;
    TG_CHECK(ft == 0 || ft == 1 || ft == 5 || ft == 6 || ft == 10 || ft == 24, TG_ERR_ARG, "tg_gemm: GEGLU epilogue needs a tile with 64-column wave tiles");
    TG_CHECK(d->M > 64, TG_ERR_ARG, "tg_gemm: GEGLU epilogue needs M > 64");
  }
  const int ctot = d->c0 + (d->a1 ? d->c1 : 0);
  if (d->a1) TG_CHECK(d->c0 % BK == 0, TG_ERR_ARG, "tg_gemm: two-source A needs c0 %% 64 == 0 (c0=%d)", d->c0);
  TG_CHECK(d->a_coef == nullptr || d->mode == 1, TG_ERR_ARG, "tg_gemm: a_coef is a conv (mode 1) argument");
  if (d->mode == 1) {
    TG_CHECK(ctot % BK == 0, TG_ERR_ARG, "tg_gemm conv: channels %% 64 required (c=%d)", ctot);
    TG_CHECK(d->K == 9L * ctot, TG_ERR_ARG, "tg_gemm conv: K must be 9*(c0+c1)");
    TG_CHECK(d->stride == 1 || d->stride == 2, TG_ERR_ARG, "tg_gemm conv: stride 1|2");
    TG_CHECK(!(d->upsample && d->stride != 1), TG_ERR_ARG, "tg_gemm conv: upsample needs stride 1");
    TG_CHECK(d->M == (int64_t)d->batch * d->out_h * d->out_w, TG_ERR_ARG, "tg_gemm conv: M != batch*out_h*out_w");
    TG_CHECK(d->pad_mode == 0 || (d->pad_mode == 1 && d->stride == 2 && !d->upsample), TG_ERR_ARG,
             "tg_gemm conv: pad_mode 1 (bottom / right padding) is the stride-2 encoder downsample only");
    const int pad2 = d->pad_mode == 1 ? 1 : 2;
    const int eh = d->upsample ? 2 * d->in_h : (d->in_h + pad2 - 3) / d->stride + 1;
    const int ew = d->upsample ? 2 * d->in_w : (d->in_w + pad2 - 3) / d->stride + 1;
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
  if (d->lda > 0 || d->ldw > 0) {
    TG_CHECK(d->mode == 0 && d->a1 == nullptr && d->a_rows_per_batch <= 0 && d->ln_u == nullptr, TG_ERR_ARG,
             "tg_gemm: lda / ldw belong to the plain single-source GEMM (no conv, no second source, no batched A, no LayerNorm fold)");
    TG_CHECK((d->lda <= 0 || (d->lda >= d->K && d->lda % 8 == 0)) && (d->ldw <= 0 || (d->ldw >= d->K && d->ldw % 8 == 0)), TG_ERR_ARG,
             "tg_gemm: lda = %lld / ldw = %lld must be >= K and multiples of 8", (long long)d->lda, (long long)d->ldw);
  }
  if (d->bvec) TG_CHECK(d->rows_per_batch > 0, TG_ERR_ARG, "tg_gemm: bvec needs rows_per_batch");
  if (d->ln_u != nullptr || d->ln_v != nullptr || d->ln_rows != nullptr) {
    TG_CHECK(d->ln_u && d->ln_v, TG_ERR_ARG, "tg_gemm: the LayerNorm fold needs both ln_u and ln_v");
    TG_CHECK(d->mode == 0 && d->a1 == nullptr && !d->bvec && !d->res && d->act == TG_ACT_NONE && d->force_split_k <= 1 && (d->force_tile == 0 || d->force_tile == 24 || d->force_tile == 25),
             TG_ERR_ARG, "tg_gemm: the LayerNorm fold takes a plain single-source GEMM with a linear or GEGLU epilogue (no residual / per-batch vector / split)");
    TG_CHECK(d->K % 32 == 0 && d->N % 8 == 0 && d->ln_eps > 0.f, TG_ERR_ARG, "tg_gemm: the LayerNorm fold needs K %% 32 == 0, N %% 8 == 0, eps > 0");
    TG_CHECK((reinterpret_cast<uintptr_t>(d->ln_u) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->ln_v) & 15) == 0 &&
             (reinterpret_cast<uintptr_t>(d->ln_rows) & 7) == 0, TG_ERR_ARG, "tg_gemm: ln_u / ln_v must be 16-byte, ln_rows 8-byte aligned");
  }
  return TG_OK;
}

}  // namespace

extern "C" int tg_gemm_plan(const tg_gemm_desc* d, int32_t* tile_m, int32_t* tile_n, int32_t* splits, int32_t* kernel_kind) {
  int rc = validate(d);
  if (rc != TG_OK) return rc;
  if (pp_selected(d) || pp160_selected(d)) {
    if (tile_m) *tile_m = 256;
    if (tile_n) *tile_n = pp_selected(d) ? 256 : 160;
    if (splits) *splits = 1;
    if (kernel_kind) *kernel_kind = 7;
    return TG_OK;
  }
  if (d->ln_u != nullptr) {
    if (tile_m) *tile_m = 128;
    if (tile_n) *tile_n = ln_t160_of(d) ? 160 : 128;
    if (splits) *splits = 1;
    if (kernel_kind) *kernel_kind = 6;
    return TG_OK;
  }
  if (const int sp = slab_splits_of(d); sp > 0) {
    if (tile_m) *tile_m = 128;
    if (tile_n) *tile_n = 320;
    if (splits) *splits = sp;
    if (kernel_kind) *kernel_kind = 4;
    return TG_OK;
  }
  if (const int bt = bt_tile_of(d); bt >= 0) {
    if (tile_m) *tile_m = 256;
    if (tile_n) *tile_n = 256;
    if (splits) *splits = 1;
    if (kernel_kind) *kernel_kind = 3;
    return TG_OK;
  }
  Plan pl = make_plan(d);
  if (tile_m) *tile_m = kTiles[pl.tile].bm;
  if (tile_n) *tile_n = kTiles[pl.tile].bn;
  if (splits) *splits = pl.s;
  if (kernel_kind) *kernel_kind = pl.halo ? 2 : (d->mode == 1 ? 1 : 0);
  return TG_OK;
}

extern "C" int tg_gemm_gn_partial_blocks(const tg_gemm_desc* d) {
  if (validate(d) != TG_OK) return 0;
  if (pp_selected(d) || pp160_selected(d) || d->ln_u != nullptr) return 0;
  return gn_partial_blocks_of(d);
}

extern "C" int64_t tg_gemm_workspace_bytes(const tg_gemm_desc* d) {
  if (validate(d) != TG_OK) return -1;
  if (pp_selected(d) || pp160_selected(d) || d->ln_u != nullptr) return 0;
  if (const int sp = slab_splits_of(d); sp > 0) return sp > 1 ? (d->M / 128) * (d->N / 320) * sp * 128 * 320 * 4 : 0;
  if (bt_tile_of(d) >= 0) return 0;
  return plan_workspace_bytes(make_plan(d));
}

extern "C" int tg_gemm(const tg_gemm_desc* d, void* stream) {
  int rc = validate(d);
  if (rc != TG_OK) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == TG_BF16) return launch_gemm<bf16_t>(d, st);
  return launch_gemm<f16_t>(d, st);
}

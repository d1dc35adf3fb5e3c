// Per-box cross-attention guidance reductions (reference utils/guidance.py:91-148, 223-233) as wavefront
// primitives: one wave owns one attention head; the masked top-k MEAN is computed without sorting by a
// 32-step radix select on the (non-negative) float bit pattern using __ballot/__popcll counts, then one
// wave_sum.  Deterministic: heads are accumulated in head order by a single lane (no float atomics).
#include "tg_common.h"

namespace {

constexpr int G_WAVES = 4;

__device__ __forceinline__ unsigned long long lanemask_lt(int lane) { return lane == 0 ? 0ull : (~0ull >> (64 - lane)); }

// k-th largest of x[0..n) (all x >= 0), values in LDS; returns threshold bits; also count(x > thr) and sum(x > thr)
__device__ void wave_topk_select(const float* x, int n, int k, int lane, float& thr, int& cnt_gt, float& sum_gt) {
  unsigned int cand = 0u;
  for (int bit = 30; bit >= 0; --bit) {   // sign bit is always 0
    const unsigned int trial = cand | (1u << bit);
    int c = 0;
    for (int i = lane; i - lane < n; i += 64) {
      const bool ge = i < n && __float_as_uint(x[i]) >= trial;
      c += __popcll(__ballot(ge));
    }
    if (c >= k) cand = trial;   // at least k values >= trial: the k-th largest is >= trial
  }
  thr = __uint_as_float(cand);
  int c = 0;
  float s = 0.f;
  for (int i = lane; i - lane < n; i += 64) {
    const bool gt = i < n && x[i] > thr;
    c += __popcll(__ballot(gt));
    s += gt ? x[i] : 0.f;
  }
  cnt_gt = c;
  sum_gt = wave_sum(s);
}

// each *_block function is executed by a whole 256-thread block and returns (in thread 0) the term the reference adds to
// the loss for ONE (attention map, object, token position); `grad` (optional) is accumulated in place
__device__ float guidance_topk_block(float* sh, const float* attn, int heads, int hw, int n_tok, int token, const float* mask,
                                     int k_fg, int k_bg, float fg_w, float bg_w, float scale, float* grad, int h0 = 0, int hcnt = 1 << 30, float* head_out = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* xf = sh + (size_t)wave * 2 * hw;   // A * M
  float* xb = xf + hw;                      // A * (1 - M)
  float* head_loss = sh + (size_t)G_WAVES * 2 * hw;  // [heads]
  const int h_end = heads < h0 + hcnt ? heads : h0 + hcnt;
  if (head_out != nullptr) head_loss = head_out;      // per-head terms go to the caller's array (folded later, in the same order)
  for (int h = h0 + wave; h < h_end; h += G_WAVES) {
    const float* col = attn + (long)h * hw * n_tok + token;
    for (int i = lane; i < hw; i += 64) {
      const float a = col[(long)i * n_tok], m = mask[i];
      xf[i] = a * m;
      xb[i] = a * (1.f - m);
    }
    __builtin_amdgcn_wave_barrier();
    float thr_f, thr_b, sf, sb;
    int cf, cb;
    wave_topk_select(xf, hw, k_fg, lane, thr_f, cf, sf);
    wave_topk_select(xb, hw, k_bg, lane, thr_b, cb, sb);
    const float mean_f = (sf + (float)(k_fg - cf) * thr_f) / (float)k_fg;
    const float mean_b = (sb + (float)(k_bg - cb) * thr_b) / (float)k_bg;
    if (lane == 0) head_loss[h] = fg_w * (1.f - mean_f) + bg_w * mean_b;
    if (grad) {
      float* gcol = grad + (long)h * hw * n_tok + token;
      int run_f = 0, run_b = 0;
      const int need_f = k_fg - cf, need_b = k_bg - cb;
      for (int i = lane; i - lane < hw; i += 64) {
        const bool in = i < hw;
        const float vf = in ? xf[i] : -1.f, vb = in ? xb[i] : -1.f;
        const bool eqf = in && vf == thr_f, eqb = in && vb == thr_b;
        const unsigned long long bf = __ballot(eqf), bb = __ballot(eqb);
        const bool self = (vf > thr_f) || (eqf && run_f + __popcll(bf & lanemask_lt(lane)) < need_f);
        const bool selb = (vb > thr_b) || (eqb && run_b + __popcll(bb & lanemask_lt(lane)) < need_b);
        run_f += __popcll(bf);
        run_b += __popcll(bb);
        if (in) {
          const float m = mask[i];
          float g = 0.f;
          if (self) g -= scale * fg_w * m / (float)k_fg;
          if (selb) g += scale * bg_w * (1.f - m) / (float)k_bg;
          if (g != 0.f) gcol[(long)i * n_tok] += g;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  float term = 0.f;
  if (threadIdx.x == 0 && head_out == nullptr) {
    float s = 0.f;
    for (int h = 0; h < heads; ++h) s += head_loss[h];
    term = scale * s;
  }
  return term;
}

__device__ float guidance_ratio_block(float* head_loss, const float* attn, int heads, int hw, int n_tok, int token,
                                      const float* mask, float scale, float* grad, int h0 = 0, int hcnt = 1 << 30, float* head_out = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h_end = heads < h0 + hcnt ? heads : h0 + hcnt;
  if (head_out != nullptr) head_loss = head_out;      // per-head terms go to the caller's array (folded later, in the same order)
  for (int h = h0 + wave; h < h_end; h += G_WAVES) {
    const float* col = attn + (long)h * hw * n_tok + token;
    float sm = 0.f, sa = 0.f;
    for (int i = lane; i < hw; i += 64) {
      const float a = col[(long)i * n_tok];
      sm += a * mask[i];
      sa += a;
    }
    sm = wave_sum(sm);
    sa = wave_sum(sa);
    const float r = sm / sa;
    if (lane == 0) head_loss[h] = (1.f - r) * (1.f - r);
    if (grad) {
      float* gcol = grad + (long)h * hw * n_tok + token;
      const float c = scale / (float)heads * (-2.f) * (1.f - r) / (sa * sa);
      for (int i = lane; i < hw; i += 64) gcol[(long)i * n_tok] += c * (mask[i] * sa - sm);
    }
  }
  __syncthreads();
  float term = 0.f;
  if (threadIdx.x == 0 && head_out == nullptr) {
    float s = 0.f;
    for (int h = 0; h < heads; ++h) s += head_loss[h];
    term = scale * s / (float)heads;
  }
  return term;
}

// attention-transfer term (utils/guidance.py:223-233): per head, the masked current column and the masked reference
// column are each normalised by (their sum + eps); loss = mean over heads of the L1 distance.  ref: [heads, hw].
__device__ float guidance_ref_block(float* head_loss, const float* attn, int heads, int hw, int n_tok, int token, const float* ref,
                                    const float* mask, float eps, float scale, float* grad, int h0 = 0, int hcnt = 1 << 30, float* head_out = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h_end = heads < h0 + hcnt ? heads : h0 + hcnt;
  if (head_out != nullptr) head_loss = head_out;      // per-head terms go to the caller's array (folded later, in the same order)
  for (int h = h0 + wave; h < h_end; h += G_WAVES) {
    const float* col = attn + (long)h * hw * n_tok + token;
    const float* rcol = ref + (long)h * hw;
    float cs = 0.f, rs = 0.f;
    for (int i = lane; i < hw; i += 64) {
      const float m = mask[i];
      cs += col[(long)i * n_tok] * m;
      rs += rcol[i] * m;
    }
    const float ci = 1.f / (wave_sum(cs) + eps);
    const float ri = 1.f / (wave_sum(rs) + eps);
    float l1 = 0.f, sc = 0.f;                          // sc = sum_i sign(d_i) * cm_i  (for the gradient)
    for (int i = lane; i < hw; i += 64) {
      const float m = mask[i];
      const float cm = col[(long)i * n_tok] * m * ci;
      const float d = cm - rcol[i] * m * ri;
      l1 += fabsf(d);
      sc += (d > 0.f ? cm : (d < 0.f ? -cm : 0.f));
    }
    l1 = wave_sum(l1);
    sc = wave_sum(sc);
    if (lane == 0) head_loss[h] = l1;
    if (grad) {
      float* gcol = grad + (long)h * hw * n_tok + token;
      const float c = scale / (float)heads * ci;
      for (int i = lane; i < hw; i += 64) {
        const float m = mask[i];
        const float d = col[(long)i * n_tok] * m * ci - rcol[i] * m * ri;
        const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        gcol[(long)i * n_tok] += c * m * (sg - sc);
      }
    }
  }
  __syncthreads();
  float term = 0.f;
  if (threadIdx.x == 0 && head_out == nullptr) {
    float s = 0.f;
    for (int h = 0; h < heads; ++h) s += head_loss[h];
    term = scale * s / (float)heads;
  }
  return term;
}

__global__ __launch_bounds__(64 * G_WAVES) void guidance_topk_kernel(const float* attn, int heads, int hw, int n_tok, int token,
                                                                     const float* mask, int k_fg, int k_bg, float fg_w,
                                                                     float bg_w, float scale, float* out, float* grad) {
  extern __shared__ float sh[];
  const float t = guidance_topk_block(sh, attn, heads, hw, n_tok, token, mask, k_fg, k_bg, fg_w, bg_w, scale, grad);
  if (threadIdx.x == 0) out[0] += t;
}
__global__ __launch_bounds__(64 * G_WAVES) void guidance_ratio_kernel(const float* attn, int heads, int hw, int n_tok, int token,
                                                                      const float* mask, float scale, float* out, float* grad) {
  extern __shared__ float sh[];
  const float t = guidance_ratio_block(sh, attn, heads, hw, n_tok, token, mask, scale, grad);
  if (threadIdx.x == 0) out[0] += t;
}
__global__ __launch_bounds__(64 * G_WAVES) void guidance_ref_kernel(const float* attn, int heads, int hw, int n_tok, int token,
                                                                    const float* ref, const float* mask, float eps,
                                                                    float scale, float* out, float* grad) {
  extern __shared__ float sh[];
  const float t = guidance_ref_block(sh, attn, heads, hw, n_tok, token, ref, mask, eps, scale, grad);
  if (threadIdx.x == 0) out[0] += t;
}

// ONE launch for a whole compute_ca_lossv3 call: block i evaluates items[i] (any mix of the three term kinds, any map
// size) and writes its term to partials[i]; the last line of the loss — the sum over items — is a second tiny kernel that
// adds the partials in ITEM ORDER, i.e. exactly the sequence of `loss += term` the per-item launches perform: same bits,
// ~40 dependent launches fewer for 4 boxes x 4 keys.  Items of one launch never share a (grad, token) column (host rule).
__global__ __launch_bounds__(64 * G_WAVES) void guidance_batch_kernel(const tg_guidance_item* items, float* partials) {
  extern __shared__ float sh[];
  const tg_guidance_item it = items[blockIdx.x];
  float t;
  if (it.kind == 0) t = guidance_topk_block(sh, it.attn, it.heads, it.hw, it.n_tok, it.token, it.mask, it.k_fg, it.k_bg, it.fg_w, it.bg_w, it.scale, it.grad);
  else if (it.kind == 1) t = guidance_ratio_block(sh, it.attn, it.heads, it.hw, it.n_tok, it.token, it.mask, it.scale, it.grad);
  else t = guidance_ref_block(sh, it.attn, it.heads, it.hw, it.n_tok, it.token, it.ref, it.mask, it.eps, it.scale, it.grad);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
__global__ void guidance_fold_kernel(const float* partials, int n, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = out[0];
    for (int i = 0; i < n; ++i) s += partials[i];
    out[0] = s;
  }
}

// ---- plan form of the same launch (round 3): the item table holds SLOT INDICES instead of pointers, the pointers of the call (the
// saved maps, their gradients, the box masks, reference columns) travel by value in the kernel arguments.  A table therefore depends
// only on (boxes, token positions, keys, map shapes, loss options): the host builds it once, keeps it on the device and every later
// call is two launches with no host -> device copy — and can be captured in the hipGraph of the denoising step.  Work split: one
// block per (item, group of G_WAVES heads), one head per wave: 4 keys x 4 boxes x 20 heads = 560 waves in flight instead of 28 blocks
// walking 5 heads each.  The fold adds the per-head terms of an item in head order and the items in item order: the sequence of
// additions of the per-item launches, i.e. the same bits.
struct GuidanceSlots { const void* p[TG_GUIDANCE_MAX_SLOTS]; };

__global__ __launch_bounds__(64 * G_WAVES) void guidance_plan_kernel(const tg_guidance_pitem* items, GuidanceSlots slots, int hgroups,
                                                                     int max_heads, float* head_terms) {
  extern __shared__ float sh[];
  const int item = blockIdx.x / hgroups, h0 = (blockIdx.x - item * hgroups) * G_WAVES;
  const tg_guidance_pitem it = items[item];
  if (h0 >= it.heads) return;
  const float* attn = reinterpret_cast<const float*>(slots.p[it.attn_slot]);
  float* grad = it.grad_slot >= 0 ? reinterpret_cast<float*>(const_cast<void*>(slots.p[it.grad_slot])) : nullptr;
  const float* mask = reinterpret_cast<const float*>(slots.p[it.mask_slot]);
  const float* ref = it.ref_slot >= 0 ? reinterpret_cast<const float*>(slots.p[it.ref_slot]) : nullptr;
  float* out = head_terms + (long)item * max_heads;
  if (it.kind == 0) guidance_topk_block(sh, attn, it.heads, it.hw, it.n_tok, it.token, mask, it.k_fg, it.k_bg, it.fg_w, it.bg_w, it.scale, grad, h0, G_WAVES, out);
  else if (it.kind == 1) guidance_ratio_block(sh, attn, it.heads, it.hw, it.n_tok, it.token, mask, it.scale, grad, h0, G_WAVES, out);
  else guidance_ref_block(sh, attn, it.heads, it.hw, it.n_tok, it.token, ref, mask, it.eps, it.scale, grad, h0, G_WAVES, out);
}
// items [begin, end) of one launch group; terms are added to out[0] in item order (thread 0), after every thread folded one item's heads
__global__ __launch_bounds__(256) void guidance_plan_fold_kernel(const tg_guidance_pitem* items, int n_items, int max_heads,
                                                                 const float* head_terms, float* out) {
  __shared__ float terms[256];
  for (int base = 0; base < n_items; base += 256) {
    const int i = base + threadIdx.x;
    if (i < n_items) {
      const tg_guidance_pitem it = items[i];
      float sacc = 0.f;
      for (int h = 0; h < it.heads; ++h) sacc += head_terms[(long)i * max_heads + h];
      terms[threadIdx.x] = it.kind == 0 ? it.scale * sacc : it.scale * sacc / (float)it.heads;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = out[0];
      const int n = n_items - base < 256 ? n_items - base : 256;
      for (int j = 0; j < n; ++j) t += terms[j];
      out[0] = t;
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int tg_guidance_plan_run(const tg_guidance_pitem* items_device, int32_t n_items, int32_t max_hw_topk, int32_t max_heads,
                                    const void* const* slots_host, int32_t n_slots, float* head_terms, float* out, void* stream) {
  TG_CHECK(items_device && head_terms && out && slots_host && n_items > 0 && max_heads > 0 && max_hw_topk >= 0, TG_ERR_ARG, "tg_guidance_plan_run: bad args");
  TG_CHECK(n_slots > 0 && n_slots <= TG_GUIDANCE_MAX_SLOTS, TG_ERR_ARG, "tg_guidance_plan_run: %d pointer slots (1..%d)", n_slots, TG_GUIDANCE_MAX_SLOTS);
  const size_t lds = ((size_t)G_WAVES * 2 * max_hw_topk + max_heads) * sizeof(float);
  TG_CHECK(lds <= 160 * 1024, TG_ERR_ARG, "tg_guidance_plan_run: attention map too large for the top-k select (hw = %d needs %zu bytes of LDS, "
           "160 KB available)", max_hw_topk, lds);
  GuidanceSlots sl;
  for (int i = 0; i < TG_GUIDANCE_MAX_SLOTS; ++i) sl.p[i] = i < n_slots ? slots_host[i] : nullptr;
  for (int i = 0; i < n_slots; ++i) TG_CHECK(sl.p[i] != nullptr, TG_ERR_ARG, "tg_guidance_plan_run: slot %d is NULL", i);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (lds > 64 * 1024)
    hipFuncSetAttribute(reinterpret_cast<const void*>(guidance_plan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int hgroups = (max_heads + G_WAVES - 1) / G_WAVES;
  hipLaunchKernelGGL(guidance_plan_kernel, dim3((unsigned)(n_items * hgroups)), dim3(64 * G_WAVES), lds, st, items_device, sl, hgroups, max_heads, head_terms);
  TG_LAUNCH_CHECK();
  hipLaunchKernelGGL(guidance_plan_fold_kernel, dim3(1), dim3(256), 0, st, items_device, n_items, max_heads, head_terms, out);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_guidance_batch(const tg_guidance_item* items_device, int32_t n_items, int32_t max_hw_topk, int32_t max_heads,
                                 float* partials, float* out, void* stream) {
  TG_CHECK(items_device && partials && out && n_items > 0 && max_heads > 0 && max_hw_topk >= 0, TG_ERR_ARG, "tg_guidance_batch: bad args");
  const size_t lds = ((size_t)G_WAVES * 2 * max_hw_topk + max_heads) * sizeof(float);
  TG_CHECK(lds <= 160 * 1024, TG_ERR_ARG, "tg_guidance_batch: attention map too large for the top-k select (hw = %d needs %zu bytes of LDS, "
           "160 KB available)", max_hw_topk, lds);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (lds > 64 * 1024)
    hipFuncSetAttribute(reinterpret_cast<const void*>(guidance_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(guidance_batch_kernel, dim3((unsigned)n_items), dim3(64 * G_WAVES), lds, st, items_device, partials);
  TG_LAUNCH_CHECK();
  hipLaunchKernelGGL(guidance_fold_kernel, dim3(1), dim3(64), 0, st, partials, n_items, out);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_guidance_ref(const float* attn, int32_t heads, int32_t hw, int32_t n_tok, int32_t token,
                               const float* ref, const float* mask, float eps, float scale, float* out, float* grad,
                               void* stream) {
  TG_CHECK(attn && ref && mask && out && heads > 0 && hw > 0 && n_tok > 0 && token >= 0 && token < n_tok, TG_ERR_ARG,
           "tg_guidance_ref: bad args");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(guidance_ref_kernel, dim3(1), dim3(64 * G_WAVES), heads * sizeof(float), st, attn, heads, hw, n_tok,
                     token, ref, mask, eps, scale, out, grad);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_guidance_topk(const float* attn, int32_t heads, int32_t hw, int32_t n_tok, int32_t token,
                                const float* mask, int32_t k_fg, int32_t k_bg, float fg_w, float bg_w, float scale,
                                float* out, float* grad, void* stream) {
  TG_CHECK(attn && mask && out && heads > 0 && hw > 0 && n_tok > 0 && token >= 0 && token < n_tok, TG_ERR_ARG,
           "tg_guidance_topk: bad args");
  TG_CHECK(k_fg >= 1 && k_fg <= hw && k_bg >= 1 && k_bg <= hw, TG_ERR_ARG, "tg_guidance_topk: k out of range");
  const size_t lds = ((size_t)G_WAVES * 2 * hw + heads) * sizeof(float);
  TG_CHECK(lds <= 160 * 1024, TG_ERR_ARG, "tg_guidance_topk: attention map too large for the top-k select (hw = %d needs %zu bytes of LDS, 160 KB available)", hw, lds);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (lds > 64 * 1024)
    hipFuncSetAttribute(reinterpret_cast<const void*>(guidance_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(guidance_topk_kernel, dim3(1), dim3(64 * G_WAVES), lds, st, attn, heads, hw, n_tok, token, mask, k_fg,
                     k_bg, fg_w, bg_w, scale, out, grad);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

extern "C" int tg_guidance_ratio(const float* attn, int32_t heads, int32_t hw, int32_t n_tok, int32_t token,
                                 const float* mask, float scale, float* out, float* grad, void* stream) {
  TG_CHECK(attn && mask && out && heads > 0 && hw > 0 && n_tok > 0 && token >= 0 && token < n_tok, TG_ERR_ARG,
           "tg_guidance_ratio: bad args");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(guidance_ratio_kernel, dim3(1), dim3(64 * G_WAVES), heads * sizeof(float), st, attn, heads, hw, n_tok,
                     token, mask, scale, out, grad);
  TG_LAUNCH_CHECK();
  return TG_OK;
}

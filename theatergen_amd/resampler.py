"""IP-Adapter image projections on HIP: Perceiver ``Resampler`` (IP-Adapter-Plus), ``ImageProjModel``, ``MLPProjModel``.

Mirrors reference ``ip_adapter/resampler.py`` (``FeedForward`` :13-20, ``PerceiverAttention`` :34-78, ``Resampler``
:81-147, ``masked_mean`` :150-158) and ``ip_adapter/ip_adapter.py`` (``ImageProjModel`` :30-47, ``MLPProjModel``
:50-64): same constructor arguments and parameter names (``layers.{i}.0.to_kv.weight`` ...), so the
``image_proj`` part of an IP-Adapter checkpoint loads unchanged (``ip_adapter.py:138``).

Latent attention = the same fused MFMA kernel as the UNet (``tg_attention``): queries = the 16 latents, keys =
[image tokens ; latents] (273 keys), scale d^-0.5 = (d^-0.25)^2 applied to the scores in fp32 — the reference
scales q and k separately by d^-0.25 "for fp16 stability" and soft-maxes in fp32 (:71-73); here scores are
accumulated in fp32 by the MFMA so the single scale is exact.  The K/V input concat (:68) is not materialised
twice: LN(x) and LN(latents) are written into one [b, n1+n2, D] buffer by the LayerNorm kernels.
"""
import torch
import torch.nn as nn

import os

from . import ops
from .attention_processor import tensor_version
from .weights_pack import pack_ln_linear, skinny_pack


def _round8(n):
    return (n + 7) // 8 * 8


class PerceiverAttention(nn.Module):
    def __init__(self, *, dim, dim_head=64, heads=8):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.dim_head, self.heads = dim_head, heads
        inner = dim_head * heads
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)

    def packed(self):
        """kernel-layout weights of the skinny path, rebuilt when the parameters change: to_kv with norm1 folded (image-token rows, tg_gemm), [to_q ; to_kv] with
        norm2 folded and to_out in tg_skinny_gemm's fragment order"""
        ps = [self.norm1.weight, self.norm1.bias, self.norm2.weight, self.norm2.bias, self.to_q.weight, self.to_kv.weight, self.to_out.weight]
        key = tuple((t.data_ptr(), tensor_version(t), t.dtype, t.device) for t in ps)
        hit = getattr(self, "_pk", None)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                wx, ux, vx = pack_ln_linear(self.to_kv.weight, None, self.norm1.weight, self.norm1.bias)
                wl, ul, vl = pack_ln_linear(torch.cat([self.to_q.weight, self.to_kv.weight], dim=0), None, self.norm2.weight, self.norm2.bias)
                hit = (key, dict(kv_x=(wx, ux, vx), qkv_lat=(skinny_pack(wl), ul, vl), out=skinny_pack(self.to_out.weight)))
            self._pk = hit
        return hit[1]

    def kv_image(self, xp_pad, b, L, k, vt, ldt):
        """K rows / V^T columns of the image tokens (norm1 folded into to_kv; independent of the latents: all layers' calls are issued up front).  ``xp_pad`` is
        [b * L, D] with the latents' slots as padding rows, so the output rows are the attention's K rows one to one; the latents' slots are overwritten by
        ``run_skinny``."""
        wx, ux, vx = self.packed()["kv_x"]
        inner = self.dim_head * self.heads
        ops.gemm(xp_pad, wx, b * L, 2 * inner, xp_pad.shape[1], rows_per_batch=L, out=k, n_split=inner, out_t=vt, ldt=ldt, ln=(ux, vx, self.norm1.eps))

    def run_skinny(self, lat2d, b, n1, n2, k, vt, ldt):
        """latents [b * n2, D] -> to_out(attn) + latents; K / V^T hold the image tokens' part already (``kv_image``).  Three launches: [q | k | v] of the latents
        (norm2 folded, routed into q and behind the image rows / columns of K / V^T: resampler.py:63-68 without the concat), attention, to_out + residual."""
        pk = self.packed()
        inner = self.dim_head * self.heads
        L = n1 + n2
        es = lat2d.element_size()
        q = torch.empty((b * n2, inner), dtype=lat2d.dtype, device=lat2d.device)
        wl, ul, vl = pk["qkv_lat"]
        ops.skinny_gemm(lat2d, wl, 3 * inner, ln=(ul, vl, self.norm2.eps), rows_per_batch=n2,
                        segs=[(q.data_ptr(), inner, n2 * inner, inner, 0),
                              (k.data_ptr() + n1 * inner * es, inner, L * inner, 2 * inner, 0),
                              (vt.data_ptr() + n1 * es, ldt, inner * ldt, 3 * inner, 1)])
        o = torch.empty((b * n2, inner), dtype=lat2d.dtype, device=lat2d.device)
        ops.attention(q, inner, n2 * inner, k, inner, L * inner, vt, ldt, inner * ldt, L, b, self.heads, self.dim_head, n2,
                      self.scale, o, inner, n2 * inner)
        return ops.skinny_gemm(o, pk["out"], lat2d.shape[1], res=lat2d)

    def run(self, x2d, lat2d, b, n1, n2, residual):
        """x2d [b*n1, D] image features, lat2d [b*n2, D] latents -> to_out(attn) + residual, [b*n2, D]."""
        D = x2d.shape[1]
        inner = self.dim_head * self.heads
        L = n1 + n2
        kv_in = torch.empty((b, L, D), dtype=x2d.dtype, device=x2d.device)
        for bi in range(b):       # LN writes straight into the concat buffer (row-pitch views, no copy)
            ops.layernorm(x2d[bi * n1:(bi + 1) * n1], self.norm1.weight, self.norm1.bias, self.norm1.eps, out=kv_in[bi, :n1])
            ops.layernorm(lat2d[bi * n2:(bi + 1) * n2], self.norm2.weight, self.norm2.bias, self.norm2.eps, out=kv_in[bi, n1:])
        # q from the normalised latents = rows [n1, L) of each batch item
        q = torch.empty((b * n2, inner), dtype=x2d.dtype, device=x2d.device)
        ops.gemm(kv_in.reshape(-1)[n1 * D:], self.to_q.weight, b * n2, inner, D, out=q, a_rows_per_batch=n2, a_batch_stride=L * D)
        ldt = _round8(L)
        k = torch.empty((b * L, inner), dtype=x2d.dtype, device=x2d.device)
        vt = torch.empty((b, inner, ldt), dtype=x2d.dtype, device=x2d.device)
        ops.gemm(kv_in.reshape(b * L, D), self.to_kv.weight, b * L, 2 * inner, D, rows_per_batch=L, out=k, n_split=inner,
                 out_t=vt, ldt=ldt)
        o = torch.empty((b * n2, inner), dtype=x2d.dtype, device=x2d.device)
        ops.attention(q, inner, n2 * inner, k, inner, L * inner, vt, ldt, inner * ldt, L, b, self.heads, self.dim_head, n2,
                      self.scale, o, inner, n2 * inner)
        return ops.linear(o, self.to_out.weight, None, res=residual)


def FeedForward(dim, mult=4):
    inner = int(dim * mult)
    return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, inner, bias=False), nn.GELU(), nn.Linear(inner, dim, bias=False))


def _ff_packed(ff):
    ps = [ff[0].weight, ff[0].bias, ff[1].weight, ff[3].weight]
    key = tuple((t.data_ptr(), tensor_version(t), t.dtype, t.device) for t in ps)
    hit = getattr(ff, "_pk", None)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            w1, u1, v1 = pack_ln_linear(ff[1].weight, None, ff[0].weight, ff[0].bias)
            hit = (key, (skinny_pack(w1), u1, v1, skinny_pack(ff[3].weight)))
        ff._pk = hit
    return hit[1]


def _ff_run_skinny(ff, x2d):
    """LayerNorm folded into the first projection (+ GELU), second projection + residual: two launches (resampler.py:13-20)"""
    w1, u1, v1, w2 = _ff_packed(ff)
    h = ops.skinny_gemm(x2d, w1, ff[1].weight.shape[0], ln=(u1, v1, ff[0].eps), act=ops.ACT_GELU)
    return ops.skinny_gemm(h, w2, x2d.shape[1], res=x2d)


def _ff_run(ff, x2d):
    h = ops.layernorm(x2d, ff[0].weight, ff[0].bias, ff[0].eps)
    h = ops.linear(h, ff[1].weight, None, act=ops.ACT_GELU)
    return ops.linear(h, ff[3].weight, None, res=x2d)


class Resampler(nn.Module):
    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768, output_dim=1024,
                 ff_mult=4, max_seq_len: int = 257, apply_pos_emb: bool = False, num_latents_mean_pooled: int = 0):
        super().__init__()
        self.pos_emb = nn.Embedding(max_seq_len, embedding_dim) if apply_pos_emb else None
        self.latents = nn.Parameter(torch.randn(1, num_queries, dim) / dim ** 0.5)
        self.proj_in = nn.Linear(embedding_dim, dim)
        self.proj_out = nn.Linear(dim, output_dim)
        self.norm_out = nn.LayerNorm(output_dim)
        self.num_latents_mean_pooled = num_latents_mean_pooled
        self.to_latents_from_mean_pooled_seq = (
            nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, dim * num_latents_mean_pooled))
            if num_latents_mean_pooled > 0 else None)
        self.layers = nn.ModuleList([nn.ModuleList([PerceiverAttention(dim=dim, dim_head=dim_head, heads=heads),
                                                    FeedForward(dim=dim, mult=ff_mult)]) for _ in range(depth)])
        for p in self.parameters():
            p.requires_grad_(False)
        self._ones = None

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("theatergen_amd Resampler runs on the GPU only (no CPU fallback)")
        dt = self.proj_in.weight.dtype
        x = x.to(dt).contiguous()
        b, n1, E = x.shape
        x2d = x.reshape(b * n1, E)
        if self.pos_emb is not None:
            pe = self.pos_emb.weight[:n1]
            x2d = torch.cat([ops.add(x2d[i * n1:(i + 1) * n1], pe) for i in range(b)], dim=0)
        D = self.proj_in.weight.shape[0]
        nq = self.latents.shape[1]
        if self._skinny_ok(b, n1, D):
            return self._forward_skinny(x, b, n1, E, D, nq)
        xp = ops.linear(x2d, self.proj_in.weight, self.proj_in.bias)                    # [b*n1, D]
        n2 = nq + self.num_latents_mean_pooled
        lat = torch.empty((b, n2, D), dtype=dt, device=x.device)
        lat[:, self.num_latents_mean_pooled:].copy_(self.latents.to(dt).expand(b, nq, D))  # placement only
        if self.to_latents_from_mean_pooled_seq is not None:
            # masked_mean with an all-true mask (:133) = mean over the sequence = (1/n1) * ones^T x : a GEMM per batch item
            ones = torch.full((8, _round8(n1)), 0.0, dtype=dt, device=x.device)
            ones[0, :n1] = 1.0
            xt = ops.transpose(xp, b, n1, D)                                           # [b, D, n1]
            pad = _round8(n1)
            mean = torch.empty((b, D), dtype=dt, device=x.device)
            for bi in range(b):
                xrow = torch.zeros((D, pad), dtype=dt, device=x.device)
                xrow[:, :n1].copy_(xt[bi])
                m8 = ops.gemm(ones, xrow, 8, D, pad, out_scale=1.0 / n1)                # row 0 = mean
                mean[bi].copy_(m8[0])
            ln, lin = self.to_latents_from_mean_pooled_seq[0], self.to_latents_from_mean_pooled_seq[1]
            mp = ops.linear(ops.layernorm(mean, ln.weight, ln.bias, ln.eps), lin.weight, lin.bias)
            lat[:, :self.num_latents_mean_pooled].copy_(mp.reshape(b, self.num_latents_mean_pooled, D))
        lat2d = lat.reshape(b * n2, D)
        for attn, ff in self.layers:
            lat2d = attn.run(xp, lat2d, b, n1, n2, lat2d)
            lat2d = _ff_run(ff, lat2d)
        out = ops.linear(lat2d, self.proj_out.weight, self.proj_out.bias)
        out = ops.layernorm(out, self.norm_out.weight, self.norm_out.bias, self.norm_out.eps)
        return out.reshape(b, n2, -1)

    __call__ = forward

    def _skinny_ok(self, b, n1, D):
        """the IP-Adapter-Plus configurations (no positional embedding, no mean-pooled latents: ip_adapter.py:326-337): latent path on tg_skinny_gemm"""
        if os.environ.get("TG_RESAMPLER_SKINNY", "1") == "0" or self.pos_emb is not None or self.num_latents_mean_pooled:
            return False
        a0, f0 = self.layers[0][0], self.layers[0][1]
        inner, hidden, odim = a0.dim_head * a0.heads, f0[1].weight.shape[0], self.proj_out.weight.shape[0]
        return (D % 64 == 0 and (D // 64 in (1, 2, 3, 5, 6, 8, 10, 12, 16, 20) or (D % 128 == 0 and D // 128 in (1, 2, 3, 5, 6, 8, 10, 12, 16, 20))) and inner % 64 == 0 and hidden % 64 == 0 and odim % 32 == 0
                and self.proj_in.weight.shape[1] % 8 == 0)

    def _forward_skinny(self, x, b, n1, E, D, nq):
        """Same function as ``forward``'s generic path (tests compare the two), 28 launches instead of 48:
        * image tokens: proj_in over a [b, n1 + nq, E] buffer whose last nq rows per item are padding, then per layer ONE LayerNorm-folded to_kv GEMM writing K rows /
          V^T columns in place (independent of the latents: issued up front);
        * latents (b x nq <= a few times 32 rows): tg_skinny_gemm for every projection — norm2 + [q | k | v] routed behind the image rows, to_out + residual,
          norm + FeedForward in two launches, proj_out."""
        dt = self.proj_in.weight.dtype
        dev = x.device
        L = n1 + nq
        x_pad = torch.zeros((b, L, E), dtype=dt, device=dev)       # (allocated per call: a cached buffer would tie later calls to the grad / inference mode of the first)
        x_pad[:, :n1].copy_(x)
        xp = ops.linear(x_pad.reshape(b * L, E), self.proj_in.weight, self.proj_in.bias)      # [b * L, D]; the padding rows hold the bias (finite)
        ldt = _round8(L)
        kvs = []
        for attn, _ in self.layers:
            inner = attn.dim_head * attn.heads
            kvs.append((torch.empty((b * L, inner), dtype=dt, device=dev), torch.empty((b, inner, ldt), dtype=dt, device=dev)))
        # the image tokens' K / V^T of layer 0 on this stream, of the later layers on a side stream UNDER the first layer's latent chain (they only need xp)
        cur = torch.cuda.current_stream(dev)
        side = getattr(self, "_side", None)
        if side is None or side.device != dev:
            side = self._side = torch.cuda.Stream(device=dev)
        self.layers[0][0].kv_image(xp, b, L, kvs[0][0], kvs[0][1], ldt)
        use_side = len(self.layers) > 1 and os.environ.get("TG_RESAMPLER_SIDE", "1") == "1"
        if use_side:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for (attn, _), (k, vt) in zip(list(self.layers)[1:], kvs[1:]):
                    attn.kv_image(xp, b, L, k, vt, ldt)
        else:
            for (attn, _), (k, vt) in zip(list(self.layers)[1:], kvs[1:]):
                attn.kv_image(xp, b, L, k, vt, ldt)
        lat2d = self.latents.to(dt).expand(b, nq, D).reshape(b * nq, D).contiguous()
        for i, ((attn, ff), (k, vt)) in enumerate(zip(self.layers, kvs)):
            if i == 1 and use_side:
                cur.wait_stream(side)
            lat2d = attn.run_skinny(lat2d, b, n1, nq, k, vt, ldt)
            lat2d = _ff_run_skinny(ff, lat2d)
        pk = getattr(self, "_pk_out", None)
        w = self.proj_out.weight
        okey = (w.data_ptr(), tensor_version(w), w.dtype, w.device)
        if pk is None or pk[0] != okey:
            with torch.no_grad():
                pk = (okey, skinny_pack(w))
            self._pk_out = pk
        out = ops.skinny_gemm(lat2d, pk[1], w.shape[0], bias=self.proj_out.bias)
        out = ops.layernorm(out, self.norm_out.weight, self.norm_out.bias, self.norm_out.eps)
        return out.reshape(b, nq, -1)

    def graphed(self, x):
        """The same forward replayed from a hipGraph captured per (input shape, dtype, weights): one graph launch instead of ~40
        dependent kernel launches (the Resampler is launch-latency bound: 5 GMAC per image)."""
        if getattr(self, "_graphed", None) is None:
            from .graphs import GraphedCall
            w = self.proj_in.weight
            self._graphed = GraphedCall(self.forward, lambda: (w.data_ptr(), tensor_version(w), self.latents.data_ptr(), tensor_version(self.latents)))
        return self._graphed(x)


class ImageProjModel(nn.Module):
    """LN(reshape(Linear(image_embeds))) — reference ip_adapter.py:30-47."""

    def __init__(self, cross_attention_dim=1024, clip_embeddings_dim=1024, clip_extra_context_tokens=4):
        super().__init__()
        self.cross_attention_dim = cross_attention_dim
        self.clip_extra_context_tokens = clip_extra_context_tokens
        self.proj = nn.Linear(clip_embeddings_dim, clip_extra_context_tokens * cross_attention_dim)
        self.norm = nn.LayerNorm(cross_attention_dim)
        for p in self.parameters():
            p.requires_grad_(False)

    def forward(self, image_embeds):
        dt = self.proj.weight.dtype
        e = image_embeds.to(dt).contiguous().reshape(-1, self.proj.weight.shape[1])
        t = ops.linear(e, self.proj.weight, self.proj.bias)
        t = t.reshape(-1, self.cross_attention_dim)
        t = ops.layernorm(t, self.norm.weight, self.norm.bias, self.norm.eps)
        return t.reshape(-1, self.clip_extra_context_tokens, self.cross_attention_dim)

    __call__ = forward


class MLPProjModel(nn.Module):
    """Linear -> GELU -> Linear -> LN — reference ip_adapter.py:50-64."""

    def __init__(self, cross_attention_dim=1024, clip_embeddings_dim=1024):
        super().__init__()
        self.proj = nn.Sequential(nn.Linear(clip_embeddings_dim, clip_embeddings_dim), nn.GELU(),
                                  nn.Linear(clip_embeddings_dim, cross_attention_dim), nn.LayerNorm(cross_attention_dim))
        for p in self.parameters():
            p.requires_grad_(False)

    def forward(self, image_embeds):
        dt = self.proj[0].weight.dtype
        shp = image_embeds.shape
        e = image_embeds.to(dt).contiguous().reshape(-1, shp[-1])
        h = ops.linear(e, self.proj[0].weight, self.proj[0].bias, act=ops.ACT_GELU)
        h = ops.linear(h, self.proj[2].weight, self.proj[2].bias)
        h = ops.layernorm(h, self.proj[3].weight, self.proj[3].bias, self.proj[3].eps)
        return h.reshape(*shp[:-1], -1)

    __call__ = forward

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

from . import ops


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

    def graphed(self, x):
        """The same forward replayed from a hipGraph captured per (input shape, dtype, weights): one graph launch instead of ~40
        dependent kernel launches (the Resampler is launch-latency bound: 5 GMAC per image)."""
        if getattr(self, "_graphed", None) is None:
            from .graphs import GraphedCall
            w = self.proj_in.weight
            self._graphed = GraphedCall(self.forward, lambda: (w.data_ptr(), w._version, self.latents.data_ptr(), self.latents._version))
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

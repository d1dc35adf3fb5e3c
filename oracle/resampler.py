"""Oracle (test infrastructure): restatement of the IP-Adapter image projections.

Follows the reference:
  * ``ip_adapter/resampler.py``: ``FeedForward`` :13-20, ``reshape_tensor`` :23-31,
    ``PerceiverAttention.forward`` :49-78, ``Resampler.forward`` :127-147,
    ``masked_mean`` :150-158
  * ``ip_adapter/ip_adapter.py``: ``ImageProjModel.forward`` :41-47,
    ``MLPProjModel.forward`` :62-64
Pinned against the imported reference on ``tests/golden/resampler_*.npz``.

Weights: a state dict with the reference module's own key names
(``latents``, ``proj_in.weight``, ``layers.{i}.0.norm1.weight``,
``layers.{i}.0.to_q.weight``, ``layers.{i}.1.0.weight`` (LN), ``layers.{i}.1.1.weight``,
``layers.{i}.1.3.weight``, ``proj_out.*``, ``norm_out.*``, optional ``pos_emb.weight``,
``to_latents_from_mean_pooled_seq.{0,1}.*``).
"""
import math

import torch
import torch.nn.functional as F


def _ln(x, sd, prefix, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def reshape_tensor(x, heads):
    """reference resampler.py:23-31"""
    bs, length, _ = x.shape
    return x.view(bs, length, heads, -1).transpose(1, 2).reshape(bs, heads, length, -1)


def perceiver_attention(sd, p, x, latents, heads, dim_head):
    """reference resampler.py:49-78: scale d^-0.25 on q AND k (:71-72), softmax in fp32 (:73)."""
    x = _ln(x, sd, p + ".norm1")
    latents = _ln(latents, sd, p + ".norm2")
    b, l, _ = latents.shape
    q = F.linear(latents, sd[p + ".to_q.weight"])
    kv_input = torch.cat((x, latents), dim=-2)
    k, v = F.linear(kv_input, sd[p + ".to_kv.weight"]).chunk(2, dim=-1)
    q, k, v = (reshape_tensor(t, heads) for t in (q, k, v))
    scale = 1 / math.sqrt(math.sqrt(dim_head))
    weight = (q * scale) @ (k * scale).transpose(-2, -1)
    weight = torch.softmax(weight.float(), dim=-1).type(weight.dtype)
    out = weight @ v
    out = out.permute(0, 2, 1, 3).reshape(b, l, -1)
    return F.linear(out, sd[p + ".to_out.weight"])


def feed_forward(sd, p, x):
    """reference resampler.py:13-20: LN -> Linear(no bias) -> GELU(erf) -> Linear(no bias)"""
    h = _ln(x, sd, p + ".0")
    h = F.gelu(F.linear(h, sd[p + ".1.weight"]))
    return F.linear(h, sd[p + ".3.weight"])


def resampler_forward(sd, x, depth, heads, dim_head, num_latents_mean_pooled=0):
    """reference resampler.py:127-147"""
    if "pos_emb.weight" in sd:
        n = x.shape[1]
        x = x + sd["pos_emb.weight"][torch.arange(n)]
    latents = sd["latents"].repeat(x.size(0), 1, 1)
    x = F.linear(x, sd["proj_in.weight"], sd["proj_in.bias"])
    if num_latents_mean_pooled > 0:
        # masked_mean with an all-true mask == plain mean over the sequence (resampler.py:150-158)
        mean = x.sum(dim=1) / torch.tensor(float(x.shape[1])).clamp(min=1e-5)
        m = _ln(mean, sd, "to_latents_from_mean_pooled_seq.0")
        m = F.linear(m, sd["to_latents_from_mean_pooled_seq.1.weight"],
                     sd["to_latents_from_mean_pooled_seq.1.bias"])
        m = m.reshape(m.shape[0], num_latents_mean_pooled, -1)
        latents = torch.cat((m, latents), dim=-2)
    for i in range(depth):
        latents = perceiver_attention(sd, f"layers.{i}.0", x, latents, heads, dim_head) + latents
        latents = feed_forward(sd, f"layers.{i}.1", latents) + latents
    latents = F.linear(latents, sd["proj_out.weight"], sd["proj_out.bias"])
    return _ln(latents, sd, "norm_out")


def image_proj_model(sd, image_embeds, num_tokens, cross_attention_dim):
    """reference ip_adapter.py:41-47: LN(reshape(Linear(embeds)))"""
    t = F.linear(image_embeds, sd["proj.weight"], sd["proj.bias"])
    t = t.reshape(-1, num_tokens, cross_attention_dim)
    return _ln(t, sd, "norm")


def mlp_proj_model(sd, image_embeds):
    """reference ip_adapter.py:54-64: Linear -> GELU -> Linear -> LN"""
    h = F.gelu(F.linear(image_embeds, sd["proj.0.weight"], sd["proj.0.bias"]))
    h = F.linear(h, sd["proj.2.weight"], sd["proj.2.bias"])
    return _ln(h, sd, "proj.3")

"""Oracle (test infrastructure): CPU fp32 restatement of the SD UNet forward.

Structure follows the reference's (LMD-modified diffusers) UNet files, which are the
*structural specification* of the network (they do not run as shipped):
  * ``models/unet_2d_condition.py``: ctor channel plan :462-563, ``forward`` :725-1023
    (time emb :819-826, conv_in :879, down loop :902-936, ControlNet residuals :938-946,
    mid :950-976, up loop :980-1012, conv_norm_out/SiLU/conv_out :1015-1018)
  * ``models/unet_2d_blocks.py``: CrossAttnDownBlock2D :279-451, DownBlock2D :454-537,
    UNetMidBlock2DCrossAttn :155-276, CrossAttnUpBlock2D :540-713, UpBlock2D :716-797
  * ``models/transformer_2d.py``: ``Transformer2DModel.forward`` :216-370
  * ``models/attention.py``: ``BasicTransformerBlock.forward`` :156-240, ``FeedForward`` :243-292,
    ``GEGLU`` :317-338   (FeedForward/GEGLU are PINNED against the imported reference,
    ``tests/golden/ff_geglu.npz``)
  * SDXL ``text_time`` additional embedding: ``ip_adapter/unet_2d_condition.py:937-954``
  * attention processors: ``oracle/attention.py`` (pinned)

PARITY UNPINNED for the third-party ``diffusers==0.21.4`` pieces (source not under
/root/reference, package not installed): ``ResnetBlock2D``, ``Downsample2D``,
``Upsample2D``, ``Timesteps``, ``TimestepEmbedding``.  They are restated from the
documented 0.21.4 semantics:
  ResnetBlock2D: h = conv1(silu(GN32(x))); h += Linear(silu(temb))[:, :, None, None];
                 h = conv2(silu(GN32(h))); out = (shortcut(x) + h) / output_scale_factor,
                 shortcut = 1x1 conv iff cin != cout
  Downsample2D(use_conv, padding=1): conv3x3 stride 2;  Upsample2D(use_conv): nearest x2, conv3x3
  Timesteps(C, flip_sin_to_cos, freq_shift): e = exp(-ln(1e4) * arange(C/2) / (C/2 - shift));
                 emb = t * e; cat[sin, cos] (flipped to cat[cos, sin] when flip_sin_to_cos)
  TimestepEmbedding: Linear -> SiLU -> Linear

The state dict uses the diffusers parameter names (``down_blocks.0.resnets.0.norm1.weight``,
``...attentions.0.transformer_blocks.0.attn2.processor.to_k_ip.weight`` ...).
"""
import math

import torch
import torch.nn.functional as F

from . import attention as oattn


def cfg_get(cfg, name, default=None):
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


def _tuple(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def timestep_sinusoid(t, dim, flip_sin_to_cos=True, freq_shift=0.0, max_period=10000):
    """diffusers 0.21.4 ``get_timestep_embedding`` (call site models/unet_2d_condition.py:315-316, :819)."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / (half - freq_shift)
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _gn(sd, p, x, groups, eps):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def resnet_block(sd, p, x, temb, groups, eps, output_scale_factor=1.0):
    h = F.conv2d(F.silu(_gn(sd, p + ".norm1", x, groups, eps)), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = h + _lin(sd, p + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = F.conv2d(F.silu(_gn(sd, p + ".norm2", h, groups, eps)), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if (p + ".conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return (x + h) / output_scale_factor


def feed_forward_geglu(sd, p, x):
    """reference models/attention.py:243-292 + GEGLU :317-338: W2 (a * gelu(g)), [a, g] = W1 x."""
    a, g = _lin(sd, p + ".net.0.proj", x).chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", a * F.gelu(g))


def _attn_weights(sd, p):
    w = {k[len(p) + 1:]: v for k, v in sd.items() if k.startswith(p + ".")}
    out = {k: v for k, v in w.items() if not k.startswith("processor.")}
    for k, v in w.items():
        if k.startswith("processor."):
            out[k[len("processor."):]] = v
    return out


def basic_transformer_block(sd, p, x, enc, heads, ca_kwargs, ip_scale, num_tokens, cross_mode):
    """reference models/attention.py:156-240 (LN -> attn1, LN -> attn2, LN -> GEGLU FF; all residual)."""
    cap = {}
    if ca_kwargs.get("save_attn_to_dict") is not None:
        cap = dict(return_probs=True,
                   return_token_ca_only=ca_kwargs.get("return_token_ca_only"),
                   return_cond_ca_only=ca_kwargs.get("return_cond_ca_only", False))
    n1 = F.layer_norm(x, (x.shape[-1],), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    # AttnProcessor forces return_attntion_probs=False but still saves when a dict is given
    # (reference ip_adapter/attention_processor.py:371-391): self-attn maps are saved under the same
    # key and then overwritten by attn2's; we only keep the cross-attention one like the flow does.
    x = oattn.attn_processor(_attn_weights(sd, p + ".attn1"), heads, n1) + x
    n2 = F.layer_norm(x, (x.shape[-1],), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    w2 = _attn_weights(sd, p + ".attn2")
    if cross_mode == "ip":
        r = oattn.ip_attn_processor(w2, heads, n2, enc, ip_scale, num_tokens, **cap)
    elif cross_mode == "cn":
        r = oattn.cn_attn_processor(w2, heads, n2, enc, num_tokens)
    else:
        r = oattn.attn_processor(w2, heads, n2, enc, **cap)
    if cap and cross_mode != "cn":
        r, probs = r
        key = tuple(ca_kwargs["attn_key"])
        save_keys = ca_kwargs.get("save_keys")
        if save_keys is None or key in save_keys:
            ca_kwargs["save_attn_to_dict"][key] = probs
    x = r + x
    n3 = F.layer_norm(x, (x.shape[-1],), sd[p + ".norm3.weight"], sd[p + ".norm3.bias"], 1e-5)
    return feed_forward_geglu(sd, p + ".ff", n3) + x


def transformer_2d(sd, p, x, enc, heads, n_layers, use_linear, groups, ca_kwargs, ip_scale, num_tokens, cross_mode):
    """reference models/transformer_2d.py:216-370 (GN eps 1e-6 :146; conv or linear projections :286-293, :322-328)."""
    b, c, h, w = x.shape
    residual = x
    y = _gn(sd, p + ".norm", x, groups, 1e-6)
    if not use_linear:
        y = F.conv2d(y, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
        y = y.permute(0, 2, 3, 1).reshape(b, h * w, c)
    else:
        y = y.permute(0, 2, 3, 1).reshape(b, h * w, c)
        y = _lin(sd, p + ".proj_in", y)
    base_key = list(ca_kwargs.get("attn_key", []))
    for i in range(n_layers):
        ca_kwargs["attn_key"] = base_key + [i]
        y = basic_transformer_block(sd, f"{p}.transformer_blocks.{i}", y, enc, heads, ca_kwargs, ip_scale, num_tokens, cross_mode)
    if not use_linear:
        y = y.reshape(b, h, w, c).permute(0, 3, 1, 2)
        y = F.conv2d(y, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    else:
        y = _lin(sd, p + ".proj_out", y)
        y = y.reshape(b, h, w, c).permute(0, 3, 1, 2)
    return y + residual


def unet_forward(cfg, sd, sample, timestep, encoder_hidden_states, ip_scale=1.0, num_tokens=4,
                 cross_mode="ip", cross_attention_kwargs=None, added_cond_kwargs=None,
                 down_block_additional_residuals=None, mid_block_additional_residual=None,
                 return_intermediates=False):
    """CPU fp32 forward.  ``sample`` [B,4,h,w]; ``timestep`` scalar/int tensor; returns [B,4,h,w]."""
    boc = tuple(cfg_get(cfg, "block_out_channels"))
    nb = len(boc)
    down_types = tuple(cfg_get(cfg, "down_block_types"))
    up_types = tuple(cfg_get(cfg, "up_block_types"))
    lpb = _tuple(cfg_get(cfg, "layers_per_block", 2), nb)
    heads_t = _tuple(cfg_get(cfg, "attention_head_dim", 8), nb)   # used as NUMBER OF HEADS (unet_2d_blocks.py:202-205)
    tl_t = _tuple(cfg_get(cfg, "transformer_layers_per_block", 1), nb)
    use_linear = bool(cfg_get(cfg, "use_linear_projection", False))
    groups = cfg_get(cfg, "norm_num_groups", 32)
    eps = cfg_get(cfg, "norm_eps", 1e-5)
    ca_kwargs = {} if cross_attention_kwargs is None else cross_attention_kwargs
    inter = {}

    B = sample.shape[0]
    t = torch.as_tensor(timestep).reshape(-1).expand(B)
    t_emb = timestep_sinusoid(t, boc[0], cfg_get(cfg, "flip_sin_to_cos", True), cfg_get(cfg, "freq_shift", 0))
    emb = _lin(sd, "time_embedding.linear_2", F.silu(_lin(sd, "time_embedding.linear_1", t_emb)))
    if cfg_get(cfg, "addition_embed_type") == "text_time":
        # reference ip_adapter/unet_2d_condition.py:937-954
        text_embeds = added_cond_kwargs["text_embeds"]
        time_ids = added_cond_kwargs["time_ids"]
        te = timestep_sinusoid(time_ids.flatten(), cfg_get(cfg, "addition_time_embed_dim", 256), True, 0)
        te = te.reshape(text_embeds.shape[0], -1)
        add = torch.cat([text_embeds, te], dim=-1)
        emb = emb + _lin(sd, "add_embedding.linear_2", F.silu(_lin(sd, "add_embedding.linear_1", add)))
    inter["emb"] = emb

    x = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    res = [x]
    for i, bt in enumerate(down_types):
        ca_kwargs["attn_key"] = ["down", i]
        base = list(ca_kwargs["attn_key"])
        for j in range(lpb[i]):
            x = resnet_block(sd, f"down_blocks.{i}.resnets.{j}", x, emb, groups, eps)
            if bt == "CrossAttnDownBlock2D":
                ca_kwargs["attn_key"] = base + [j]
                x = transformer_2d(sd, f"down_blocks.{i}.attentions.{j}", x, encoder_hidden_states, heads_t[i],
                                   tl_t[i], use_linear, groups, ca_kwargs, ip_scale, num_tokens, cross_mode)
            res.append(x)
        if i != nb - 1:
            x = F.conv2d(x, sd[f"down_blocks.{i}.downsamplers.0.conv.weight"],
                         sd[f"down_blocks.{i}.downsamplers.0.conv.bias"], stride=2, padding=1)
            res.append(x)
    if down_block_additional_residuals is not None:       # ControlNet: models/unet_2d_condition.py:938-946
        res = [r + a for r, a in zip(res, down_block_additional_residuals)]

    ca_kwargs["attn_key"] = ["mid", 0]
    x = resnet_block(sd, "mid_block.resnets.0", x, emb, groups, eps)
    ca_kwargs["attn_key"] = ["mid", 0, 0]
    x = transformer_2d(sd, "mid_block.attentions.0", x, encoder_hidden_states, heads_t[-1], tl_t[-1],
                       use_linear, groups, ca_kwargs, ip_scale, num_tokens, cross_mode)
    x = resnet_block(sd, "mid_block.resnets.1", x, emb, groups, eps)
    if mid_block_additional_residual is not None:
        x = x + mid_block_additional_residual
    inter["mid"] = x

    rheads = tuple(reversed(heads_t))
    rtl = tuple(reversed(tl_t))
    rlpb = tuple(reversed(lpb))
    for i, bt in enumerate(up_types):
        ca_kwargs["attn_key"] = ["up", i]
        base = list(ca_kwargs["attn_key"])
        n = rlpb[i] + 1
        skips, res = res[-n:], res[:-n]
        for j in range(n):
            x = torch.cat([x, skips[-1 - j]], dim=1)      # unet_2d_blocks.py:648-651
            x = resnet_block(sd, f"up_blocks.{i}.resnets.{j}", x, emb, groups, eps)
            if bt == "CrossAttnUpBlock2D":
                ca_kwargs["attn_key"] = base + [j]
                x = transformer_2d(sd, f"up_blocks.{i}.attentions.{j}", x, encoder_hidden_states, rheads[i],
                                   rtl[i], use_linear, groups, ca_kwargs, ip_scale, num_tokens, cross_mode)
        if i != nb - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, sd[f"up_blocks.{i}.upsamplers.0.conv.weight"],
                         sd[f"up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    x = F.silu(_gn(sd, "conv_norm_out", x, groups, eps))
    x = F.conv2d(x, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)
    if return_intermediates:
        return x, inter
    return x

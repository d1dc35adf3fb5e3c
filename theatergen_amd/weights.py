"""Parameter tables for the UNet / IP-Adapter plans: names, shapes and synthetic (seeded) values.

No checkpoints exist offline, so benches and tests use seeded default-torch-style init;
real checkpoints load through the same names: the diffusers ``UNet2DConditionModel`` state-dict keys,
with the IP-Adapter K/V projections under ``...attn2.processor.to_{k,v}_ip.weight`` exactly where
``ModuleList(unet.attn_processors.values())`` puts them (reference ``ip_adapter/ip_adapter.py:139-140``).
"""
import math
from collections import OrderedDict

import torch

from .config import UNetConfig


def unet_param_shapes(cfg: UNetConfig, ip_adapter=True):
    """Ordered {name: shape} following the reference ctor ``models/unet_2d_condition.py:294-593`` and the
    block ctors in ``models/unet_2d_blocks.py`` (resnets / attentions / down|upsamplers)."""
    s = OrderedDict()
    boc = tuple(cfg.block_out_channels)
    nb = len(boc)
    ted = cfg.time_embed_dim
    heads_t = cfg.per_block(cfg.attention_head_dim)
    tl_t = cfg.per_block(cfg.transformer_layers_per_block)
    lpb = cfg.per_block(cfg.layers_per_block)
    ctx = cfg.cross_attention_dim

    def conv(p, cin, cout, k):
        s[p + ".weight"] = (cout, cin, k, k)
        s[p + ".bias"] = (cout,)

    def lin(p, cin, cout, bias=True):
        s[p + ".weight"] = (cout, cin)
        if bias:
            s[p + ".bias"] = (cout,)

    def norm(p, c):
        s[p + ".weight"] = (c,)
        s[p + ".bias"] = (c,)

    def resnet(p, cin, cout):
        norm(p + ".norm1", cin)
        conv(p + ".conv1", cin, cout, 3)
        lin(p + ".time_emb_proj", ted, cout)
        norm(p + ".norm2", cout)
        conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".conv_shortcut", cin, cout, 1)

    def transformer(p, c, n_layers):
        norm(p + ".norm", c)
        if cfg.use_linear_projection:
            lin(p + ".proj_in", c, c)
        else:
            conv(p + ".proj_in", c, c, 1)
        for i in range(n_layers):
            b = f"{p}.transformer_blocks.{i}"
            norm(b + ".norm1", c)
            lin(b + ".attn1.to_q", c, c, False)
            lin(b + ".attn1.to_k", c, c, False)
            lin(b + ".attn1.to_v", c, c, False)
            lin(b + ".attn1.to_out.0", c, c)
            norm(b + ".norm2", c)
            lin(b + ".attn2.to_q", c, c, False)
            lin(b + ".attn2.to_k", ctx, c, False)
            lin(b + ".attn2.to_v", ctx, c, False)
            lin(b + ".attn2.to_out.0", c, c)
            if ip_adapter:
                lin(b + ".attn2.processor.to_k_ip", ctx, c, False)
                lin(b + ".attn2.processor.to_v_ip", ctx, c, False)
            norm(b + ".norm3", c)
            lin(b + ".ff.net.0.proj", c, 8 * c)
            lin(b + ".ff.net.2", 4 * c, c)
        if cfg.use_linear_projection:
            lin(p + ".proj_out", c, c)
        else:
            conv(p + ".proj_out", c, c, 1)

    conv("conv_in", cfg.in_channels, boc[0], 3)
    lin("time_embedding.linear_1", boc[0], ted)
    lin("time_embedding.linear_2", ted, ted)
    if cfg.addition_embed_type == "text_time":
        lin("add_embedding.linear_1", cfg.projection_class_embeddings_input_dim, ted)
        lin("add_embedding.linear_2", ted, ted)

    out_c = boc[0]
    for i, bt in enumerate(cfg.down_block_types):
        in_c, out_c = out_c, boc[i]
        for j in range(lpb[i]):
            resnet(f"down_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, out_c)
            if bt == "CrossAttnDownBlock2D":
                transformer(f"down_blocks.{i}.attentions.{j}", out_c, tl_t[i])
        if i != nb - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", out_c, out_c, 3)

    resnet("mid_block.resnets.0", boc[-1], boc[-1])
    transformer("mid_block.attentions.0", boc[-1], tl_t[-1])
    resnet("mid_block.resnets.1", boc[-1], boc[-1])

    rboc = tuple(reversed(boc))
    rtl = tuple(reversed(tl_t))
    rlpb = tuple(reversed(lpb))
    out_c = rboc[0]
    for i, bt in enumerate(cfg.up_block_types):
        prev_out, out_c = out_c, rboc[i]
        in_c = rboc[min(i + 1, nb - 1)]
        n = rlpb[i] + 1
        for j in range(n):
            skip_c = in_c if j == n - 1 else out_c
            res_in = prev_out if j == 0 else out_c
            resnet(f"up_blocks.{i}.resnets.{j}", res_in + skip_c, out_c)
            if bt == "CrossAttnUpBlock2D":
                transformer(f"up_blocks.{i}.attentions.{j}", out_c, rtl[i])
        if i != nb - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", out_c, out_c, 3)

    norm("conv_norm_out", boc[0])
    conv("conv_out", boc[0], cfg.out_channels, 3)
    return s


def _fill(shapes, seed, dtype=torch.float32, norm_jitter=0.1, device=None):
    """``device``: draw on that device with ITS generator (bench start-up: no host RNG, and every rank of a multi-GPU launch draws the same values from the same
    seed on its own GPU — nothing to broadcast); None = the host generator (tests, goldens: values unchanged)."""
    g = torch.Generator(device=device).manual_seed(seed) if device is not None else torch.Generator().manual_seed(seed)
    kwd = {"device": device} if device is not None else {}
    sd = OrderedDict()
    for name, shape in shapes.items():
        leaf = name.rsplit(".", 2)[-2] if name.count(".") else name
        is_norm = ("norm" in leaf) and len(shape) == 1
        if name == "latents":
            sd[name] = (torch.randn(shape, generator=g, **kwd) / shape[-1] ** 0.5).to(dtype)
            continue
        if is_norm:
            if name.endswith(".weight"):
                sd[name] = (1.0 + norm_jitter * torch.randn(shape, generator=g, **kwd)).to(dtype)
            else:
                sd[name] = (norm_jitter * torch.randn(shape, generator=g, **kwd)).to(dtype)
            continue
        if name.endswith(".weight"):
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
        else:  # bias: fan_in of the matching weight
            wshape = shapes.get(name[:-5] + ".weight")
            fan_in = 1
            for d in (wshape[1:] if wshape is not None else shape):
                fan_in *= d
        bound = 1.0 / math.sqrt(max(fan_in, 1))
        sd[name] = ((torch.rand(shape, generator=g, **kwd) * 2 - 1) * bound).to(dtype)
    return sd


def random_unet_state_dict(cfg: UNetConfig, seed=0, ip_adapter=True, dtype=torch.float32, device=None):
    """Seeded nn.Linear/nn.Conv2d-style init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)); norm affine jittered
    around (1, 0) so a wrong gamma/beta shows up in parity tests."""
    return _fill(unet_param_shapes(cfg, ip_adapter), seed, dtype, device=device)


def resampler_param_shapes(dim, depth, dim_head, heads, num_queries, embedding_dim, output_dim, ff_mult=4,
                           max_seq_len=257, apply_pos_emb=False, num_latents_mean_pooled=0):
    """Names of reference ``ip_adapter/resampler.py::Resampler`` (:82-125)."""
    s = OrderedDict()
    inner = dim_head * heads
    if apply_pos_emb:
        s["pos_emb.weight"] = (max_seq_len, embedding_dim)
    s["latents"] = (1, num_queries, dim)
    s["proj_in.weight"] = (dim, embedding_dim)
    s["proj_in.bias"] = (dim,)
    s["proj_out.weight"] = (output_dim, dim)
    s["proj_out.bias"] = (output_dim,)
    s["norm_out.weight"] = (output_dim,)
    s["norm_out.bias"] = (output_dim,)
    if num_latents_mean_pooled > 0:
        s["to_latents_from_mean_pooled_seq.0.weight"] = (dim,)
        s["to_latents_from_mean_pooled_seq.0.bias"] = (dim,)
        s["to_latents_from_mean_pooled_seq.1.weight"] = (dim * num_latents_mean_pooled, dim)
        s["to_latents_from_mean_pooled_seq.1.bias"] = (dim * num_latents_mean_pooled,)
    for i in range(depth):
        p = f"layers.{i}"
        s[p + ".0.norm1.weight"] = (dim,)
        s[p + ".0.norm1.bias"] = (dim,)
        s[p + ".0.norm2.weight"] = (dim,)
        s[p + ".0.norm2.bias"] = (dim,)
        s[p + ".0.to_q.weight"] = (inner, dim)
        s[p + ".0.to_kv.weight"] = (inner * 2, dim)
        s[p + ".0.to_out.weight"] = (dim, inner)
        s[p + ".1.0.weight"] = (dim,)
        s[p + ".1.0.bias"] = (dim,)
        s[p + ".1.1.weight"] = (int(dim * ff_mult), dim)
        s[p + ".1.3.weight"] = (dim, int(dim * ff_mult))
    return s


def random_resampler_state_dict(seed=0, **kw):
    shapes = resampler_param_shapes(**kw)
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, shape in shapes.items():
        is_norm = len(shape) == 1 and (".norm" in name or name.startswith("norm_out") or name.endswith(".1.0.weight")
                                       or name.endswith(".1.0.bias") or name.startswith("to_latents_from_mean_pooled_seq.0"))
        if name == "latents":
            sd[name] = torch.randn(shape, generator=g) / shape[-1] ** 0.5
        elif name == "pos_emb.weight":
            sd[name] = torch.randn(shape, generator=g)
        elif is_norm:
            sd[name] = (1.0 + 0.1 * torch.randn(shape, generator=g)) if name.endswith("weight") else 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = shape[-1] if name.endswith(".weight") else shapes[name[:-5] + ".weight"][-1]
            sd[name] = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
    return sd


def controlnet_param_shapes(cfg: UNetConfig, conditioning_channels=3, conditioning_embedding_out_channels=(16, 32, 96, 256)):
    """diffusers ``ControlNetModel`` parameter names: the UNet's encoder half + ``controlnet_cond_embedding.*`` +
    ``controlnet_down_blocks.N`` / ``controlnet_mid_block`` (1x1 convs)."""
    s = OrderedDict()
    for k, v in unet_param_shapes(cfg, ip_adapter=False).items():
        if k.startswith(("conv_in.", "time_embedding.", "down_blocks.", "mid_block.")):
            s[k] = v
    boc = tuple(cfg.block_out_channels)
    ce = tuple(conditioning_embedding_out_channels)
    p = "controlnet_cond_embedding"
    s[f"{p}.conv_in.weight"] = (ce[0], conditioning_channels, 3, 3)
    s[f"{p}.conv_in.bias"] = (ce[0],)
    for i in range(len(ce) - 1):
        s[f"{p}.blocks.{2 * i}.weight"] = (ce[i], ce[i], 3, 3)
        s[f"{p}.blocks.{2 * i}.bias"] = (ce[i],)
        s[f"{p}.blocks.{2 * i + 1}.weight"] = (ce[i + 1], ce[i], 3, 3)
        s[f"{p}.blocks.{2 * i + 1}.bias"] = (ce[i + 1],)
    s[f"{p}.conv_out.weight"] = (boc[0], ce[-1], 3, 3)
    s[f"{p}.conv_out.bias"] = (boc[0],)
    lpb = cfg.per_block(cfg.layers_per_block)
    chans = [boc[0]]
    for i in range(len(boc)):
        chans += [boc[i]] * lpb[i]
        if i != len(boc) - 1:
            chans.append(boc[i])
    for i, c in enumerate(chans):
        s[f"controlnet_down_blocks.{i}.weight"] = (c, c, 1, 1)
        s[f"controlnet_down_blocks.{i}.bias"] = (c,)
    s["controlnet_mid_block.weight"] = (boc[-1], boc[-1], 1, 1)
    s["controlnet_mid_block.bias"] = (boc[-1],)
    return s


def random_controlnet_state_dict(cfg: UNetConfig, seed=0, dtype=torch.float32, **kw):
    """Seeded init like ``random_unet_state_dict``; the zero-convs get RANDOM (non-zero) weights so that parity tests see
    the whole branch (a freshly initialised ControlNet would output exact zeros)."""
    return _fill(controlnet_param_shapes(cfg, **kw), seed, dtype)


def vae_decoder_param_shapes(cfg):
    """diffusers ``AutoencoderKL`` decoder-side parameter names (``post_quant_conv`` + ``decoder.*``)."""
    s = OrderedDict()
    boc = tuple(cfg.block_out_channels)
    lc = cfg.latent_channels
    s["post_quant_conv.weight"] = (lc, lc, 1, 1)
    s["post_quant_conv.bias"] = (lc,)
    s["decoder.conv_in.weight"] = (boc[-1], lc, 3, 3)
    s["decoder.conv_in.bias"] = (boc[-1],)

    def resnet(p, cin, cout):
        s[p + ".norm1.weight"] = (cin,)
        s[p + ".norm1.bias"] = (cin,)
        s[p + ".conv1.weight"] = (cout, cin, 3, 3)
        s[p + ".conv1.bias"] = (cout,)
        s[p + ".norm2.weight"] = (cout,)
        s[p + ".norm2.bias"] = (cout,)
        s[p + ".conv2.weight"] = (cout, cout, 3, 3)
        s[p + ".conv2.bias"] = (cout,)
        if cin != cout:
            s[p + ".conv_shortcut.weight"] = (cout, cin, 1, 1)
            s[p + ".conv_shortcut.bias"] = (cout,)

    c = boc[-1]
    resnet("decoder.mid_block.resnets.0", c, c)
    a = "decoder.mid_block.attentions.0"
    s[a + ".group_norm.weight"] = (c,)
    s[a + ".group_norm.bias"] = (c,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        s[f"{a}.{n}.weight"] = (c, c)
        s[f"{a}.{n}.bias"] = (c,)
    resnet("decoder.mid_block.resnets.1", c, c)
    prev = c
    for i, out_c in enumerate(reversed(boc)):
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else out_c, out_c)
        if i != len(boc) - 1:
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (out_c,)
        prev = out_c
    s["decoder.conv_norm_out.weight"] = (boc[0],)
    s["decoder.conv_norm_out.bias"] = (boc[0],)
    s["decoder.conv_out.weight"] = (cfg.out_channels, boc[0], 3, 3)
    s["decoder.conv_out.bias"] = (cfg.out_channels,)
    return s


def random_vae_decoder_state_dict(cfg, seed=0, dtype=torch.float32):
    return _fill(vae_decoder_param_shapes(cfg), seed, dtype)


def vae_encoder_param_shapes(cfg):
    """diffusers ``AutoencoderKL`` encoder-side parameter names (``encoder.*`` + ``quant_conv``): the SD VAE encoder has
    34 163 592 parameters (+ 72 in quant_conv)."""
    s = OrderedDict()
    boc = tuple(cfg.block_out_channels)
    lc = cfg.latent_channels
    s["encoder.conv_in.weight"] = (boc[0], cfg.in_channels, 3, 3)
    s["encoder.conv_in.bias"] = (boc[0],)

    def resnet(p, cin, cout):
        s[p + ".norm1.weight"] = (cin,)
        s[p + ".norm1.bias"] = (cin,)
        s[p + ".conv1.weight"] = (cout, cin, 3, 3)
        s[p + ".conv1.bias"] = (cout,)
        s[p + ".norm2.weight"] = (cout,)
        s[p + ".norm2.bias"] = (cout,)
        s[p + ".conv2.weight"] = (cout, cout, 3, 3)
        s[p + ".conv2.bias"] = (cout,)
        if cin != cout:
            s[p + ".conv_shortcut.weight"] = (cout, cin, 1, 1)
            s[p + ".conv_shortcut.bias"] = (cout,)

    prev = boc[0]
    for i, out_c in enumerate(boc):
        for j in range(cfg.layers_per_block):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", prev if j == 0 else out_c, out_c)
        if i != len(boc) - 1:
            s[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
            s[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (out_c,)
        prev = out_c
    c = boc[-1]
    resnet("encoder.mid_block.resnets.0", c, c)
    a = "encoder.mid_block.attentions.0"
    s[a + ".group_norm.weight"] = (c,)
    s[a + ".group_norm.bias"] = (c,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        s[f"{a}.{n}.weight"] = (c, c)
        s[f"{a}.{n}.bias"] = (c,)
    resnet("encoder.mid_block.resnets.1", c, c)
    s["encoder.conv_norm_out.weight"] = (c,)
    s["encoder.conv_norm_out.bias"] = (c,)
    s["encoder.conv_out.weight"] = (2 * lc, c, 3, 3)
    s["encoder.conv_out.bias"] = (2 * lc,)
    s["quant_conv.weight"] = (2 * lc, 2 * lc, 1, 1)
    s["quant_conv.bias"] = (2 * lc,)
    return s


def random_vae_state_dict(cfg, seed=0, dtype=torch.float32):
    """encoder + decoder (the full AutoencoderKL state dict)"""
    shapes = OrderedDict(vae_encoder_param_shapes(cfg))
    shapes.update(vae_decoder_param_shapes(cfg))
    return _fill(shapes, seed, dtype)

"""Oracle (test infrastructure): CPU fp32 restatement of ``AutoencoderKL.decode`` as the reference calls it
(``models/pipelines.py:468`` / ``:849-854``: ``image = vae.decode(latents / vae.config.scaling_factor,
return_dict=False)[0]``; the SD-1.5 VAE, ``generate.py:56-60``).

PARITY UNPINNED: ``AutoencoderKL`` / ``Decoder`` / ``UNetMidBlock2D`` / ``UpDecoderBlock2D`` live in the third-party
``diffusers==0.21.4`` (``requirements.txt:13``; source not under /root/reference, package not installed).  Restated
from the documented 0.21.4 semantics (SD VAE config: ``block_out_channels=(128, 256, 512, 512)``, ``layers_per_block=2``,
``norm_num_groups=32``, ``latent_channels=4``, ``out_channels=3``, ``scaling_factor=0.18215``, act SiLU, eps 1e-6):

  decode(z): z = post_quant_conv(z) [1x1, 4 -> 4]; Decoder:
     x = conv_in(z) [3x3, 4 -> 512]
     mid: ResnetBlock2D(512, temb=None) -> Attention(512, heads=1, dim_head=512, GroupNorm(32) on the input,
          q/k/v/out Linear WITH bias, residual connection, rescale 1) -> ResnetBlock2D(512)
     up_blocks (reversed channels 512, 512, 256, 128): 3 x ResnetBlock2D (first one changes channels, 1x1 shortcut)
          + Upsample2D (nearest x2, conv3x3) on all but the last block
     conv_norm_out GroupNorm(32) -> SiLU -> conv_out [3x3, 128 -> 3]
  ResnetBlock2D(temb=None): h = conv1(silu(GN(x))); h = conv2(silu(GN(h))); out = shortcut(x) + h   (scale factor 1)

  encode(x): Encoder(double_z=True): conv_in [3x3, 3 -> 128]; DownEncoderBlock2D x 4 (2 x ResnetBlock2D, first one changes
     channels; all but the last end in Downsample2D(use_conv=True, padding=0): F.pad(h, (0, 1, 0, 1)) then conv3x3 stride 2
     padding 0); UNetMidBlock2D as above; conv_norm_out GroupNorm(32, eps 1e-6) -> SiLU -> conv_out [3x3, 512 -> 8];
     moments = quant_conv(h) [1x1, 8 -> 8]; DiagonalGaussianDistribution: mean, logvar = chunk(moments, 2, dim=1),
     logvar clamped to [-30, 20], sample = mean + exp(0.5 logvar) * randn(generator); the reference multiplies the sample
     by scaling_factor (models/pipelines.py:157-159, 624-626).

State-dict names are diffusers' (``post_quant_conv.weight``, ``decoder.mid_block.attentions.0.to_q.weight``,
``decoder.up_blocks.1.resnets.0.conv_shortcut.weight``, ``decoder.up_blocks.0.upsamplers.0.conv.weight`` ...).
"""
import torch
import torch.nn.functional as F


def _gn(sd, p, x, groups, eps):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def resnet_block(sd, p, x, groups=32, eps=1e-6):
    h = F.conv2d(F.silu(_gn(sd, p + ".norm1", x, groups, eps)), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_gn(sd, p + ".norm2", h, groups, eps)), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if p + ".conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def mid_attention(sd, p, x, groups=32, eps=1e-6):
    b, c, h, w = x.shape
    y = _gn(sd, p + ".group_norm", x, groups, eps).reshape(b, c, h * w).transpose(1, 2)     # [b, hw, c]
    q = F.linear(y, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])
    k = F.linear(y, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])
    v = F.linear(y, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    probs = torch.softmax(torch.baddbmm(torch.empty(b, h * w, h * w), q, k.transpose(1, 2), beta=0, alpha=c ** -0.5), dim=-1)
    o = F.linear(torch.bmm(probs, v), sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return o.transpose(1, 2).reshape(b, c, h, w) + x


def decode(cfg, sd, latents, scaling_factor=None):
    """latents [B, 4, h, w] as produced by the denoising loop (NOT yet divided by the scaling factor) -> image [B, 3, 8h, 8w]"""
    boc = tuple(cfg["block_out_channels"])
    groups = cfg.get("norm_num_groups", 32)
    lpb = cfg.get("layers_per_block", 2)
    sf = cfg.get("scaling_factor", 0.18215) if scaling_factor is None else scaling_factor
    z = F.conv2d(latents / sf, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    x = F.conv2d(z, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    x = resnet_block(sd, "decoder.mid_block.resnets.0", x, groups)
    x = mid_attention(sd, "decoder.mid_block.attentions.0", x, groups)
    x = resnet_block(sd, "decoder.mid_block.resnets.1", x, groups)
    for i in range(len(boc)):
        for j in range(lpb + 1):
            x = resnet_block(sd, f"decoder.up_blocks.{i}.resnets.{j}", x, groups)
        if i != len(boc) - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"], sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    x = F.silu(_gn(sd, "decoder.conv_norm_out", x, groups, 1e-6))
    return F.conv2d(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


def encode_moments(cfg, sd, image):
    """image [B, 3, H, W] in [-1, 1] -> moments [B, 2 * latent_channels, H/8, W/8] (mean | logvar before clamping)"""
    boc = tuple(cfg["block_out_channels"])
    groups = cfg.get("norm_num_groups", 32)
    lpb = cfg.get("layers_per_block", 2)
    x = F.conv2d(image, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    for i in range(len(boc)):
        for j in range(lpb):
            x = resnet_block(sd, f"encoder.down_blocks.{i}.resnets.{j}", x, groups)
        if i != len(boc) - 1:
            x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
            x = F.conv2d(x, sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"], sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"],
                         stride=2, padding=0)
    x = resnet_block(sd, "encoder.mid_block.resnets.0", x, groups)
    x = mid_attention(sd, "encoder.mid_block.attentions.0", x, groups)
    x = resnet_block(sd, "encoder.mid_block.resnets.1", x, groups)
    x = F.silu(_gn(sd, "encoder.conv_norm_out", x, groups, 1e-6))
    x = F.conv2d(x, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    return F.conv2d(x, sd["quant_conv.weight"], sd["quant_conv.bias"])


def sample_latents(cfg, moments, noise, scaling_factor=None):
    """DiagonalGaussianDistribution.sample with the given noise, times scaling_factor (reference models/pipelines.py:157-159)"""
    sf = cfg.get("scaling_factor", 0.18215) if scaling_factor is None else scaling_factor
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return sf * (mean + torch.exp(0.5 * logvar) * noise)

"""``AutoencoderKL`` on the native kernels.

``decode``: the per-image step after the 50-step loop, ``image = vae.decode(latents / vae.config.scaling_factor,
return_dict=False)[0]`` (reference ``models/pipelines.py:468, 849-854``), so that images/s can be quoted end to end.
``encode``: ``vae.encode(image).latent_dist.sample(generator)`` of the stage-2 path (reference ``models/pipelines.py:131-160,
624-626``: the pasted mid image becomes the frozen latents, re-noised at every timestep by ``scheduler.add_noise`` :629-631).

Same module tree / state-dict names as diffusers 0.21.4 (``encoder.conv_in``, ``encoder.down_blocks.N.{resnets,
downsamplers}``, ``encoder.mid_block``, ``encoder.conv_norm_out``, ``encoder.conv_out``, ``quant_conv``; ``post_quant_conv``,
``decoder.conv_in``, ``decoder.mid_block.{resnets, attentions}``, ``decoder.up_blocks.N.{resnets, upsamplers}``,
``decoder.conv_norm_out``, ``decoder.conv_out``).  The encoder's ``Downsample2D(padding=0)`` pads bottom / right only
(``F.pad(x, (0, 1, 0, 1))`` then a stride-2 conv): ``pad_mode=1`` of the implicit-GEMM conv.  Everything runs token-major on the
UNet's kernels: GroupNorm(+SiLU), conv3x3 (LDS-halo kernel at widths <= 64, implicit GEMM above), nearest-x2 upsample
folded into the conv gather, 1x1 shortcut GEMMs with the residual in the epilogue.  The mid block's single-head d = 512
attention (4096 tokens) is outside the fused flash kernel's register budget and runs once per image: fused Q|K|V^T
projection GEMM -> scores GEMM -> ``tg_softmax_rows`` -> PV GEMM -> output projection with the residual fused.
"""
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops
from .unet import ResnetBlock2D, Upsample2D, _Act, _Packed
from .weights_pack import pack_conv3x3


class VAEConfig(SimpleNamespace):
    def get(self, k, d=None):
        return getattr(self, k, d)

    def __getitem__(self, k):
        return getattr(self, k)


def sd_vae_config(**kw):
    """stabilityai / runwayml SD-1.5 VAE (``vae/config.json``)."""
    c = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
             norm_num_groups=32, scaling_factor=0.18215, sample_size=512)
    c.update(kw)
    return VAEConfig(**c)


def tiny_vae_config():
    return sd_vae_config(block_out_channels=(64, 64, 128, 128), sample_size=64)


class VAEAttention(nn.Module):
    """diffusers ``Attention(C, heads=1, dim_head=C, bias=True, norm_num_groups=32, residual_connection=True)``"""

    def __init__(self, channels, groups=32, eps=1e-6):
        super().__init__()
        self.channels, self.groups, self.eps = channels, groups, eps
        self.group_norm = nn.GroupNorm(groups, channels, eps=eps)
        self.to_q = nn.Linear(channels, channels)
        self.to_k = nn.Linear(channels, channels)
        self.to_v = nn.Linear(channels, channels)
        self.to_out = nn.ModuleList([nn.Linear(channels, channels)])
        self._p = _Packed()

    def run(self, x: _Act):
        C, N, B = self.channels, x.hw, x.b
        if N % 8 != 0:
            raise RuntimeError("VAE attention needs a token count that is a multiple of 8")
        ws = [self.to_k.weight, self.to_v.weight, self.to_k.bias, self.to_v.bias]
        wkv, bkv = self._p.get("kv", ws, lambda: (torch.cat([t.detach() for t in ws[:2]], 0).contiguous(),
                                                  torch.cat([t.detach() for t in ws[2:]], 0).contiguous()))
        y = ops.groupnorm(x.t, B, N, self.groups, self.eps, self.group_norm.weight, self.group_norm.bias, silu=False)
        q = ops.linear(y, self.to_q.weight, self.to_q.bias)
        k = torch.empty((B * N, C), dtype=y.dtype, device=y.device)
        vt = torch.empty((B, C, N), dtype=y.dtype, device=y.device)
        ops.gemm(y, wkv, B * N, 2 * C, C, bias=bkv, rows_per_batch=N, out=k, n_split=C, out_t=vt, ldt=N)   # K rows + V^T
        o = torch.empty((B * N, C), dtype=y.dtype, device=y.device)
        scores = torch.empty((N, N), dtype=y.dtype, device=y.device)
        for b in range(B):
            ops.gemm(q[b * N:(b + 1) * N], k[b * N:(b + 1) * N], N, N, C, out=scores)        # S = Q K^T (K rows as the [N, K] operand)
            ops.softmax_rows(scores, scale=float(C) ** -0.5, out=scores)
            ops.gemm(scores, vt[b], N, C, N, out=o[b * N:(b + 1) * N])                       # O = P V  (V^T rows as the operand)
        out = ops.linear(o, self.to_out[0].weight, self.to_out[0].bias, res=x.t)
        return _Act(out, x.b, x.h, x.w, C)


class _UpBlock(nn.Module):
    def __init__(self):
        super().__init__()
        self.resnets = nn.ModuleList()
        self.upsamplers = None


class Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc = tuple(cfg.block_out_channels)
        g = cfg.norm_num_groups
        self.conv_in = nn.Conv2d(cfg.latent_channels, boc[-1], 3, padding=1)
        mid = nn.Module()
        mid.resnets = nn.ModuleList([ResnetBlock2D(boc[-1], boc[-1], None, g, 1e-6), ResnetBlock2D(boc[-1], boc[-1], None, g, 1e-6)])
        mid.attentions = nn.ModuleList([VAEAttention(boc[-1], g, 1e-6)])
        self.mid_block = mid
        self.up_blocks = nn.ModuleList()
        rboc = tuple(reversed(boc))
        out_c = rboc[0]
        for i, c in enumerate(rboc):
            prev, out_c = out_c, c
            blk = _UpBlock()
            for j in range(cfg.layers_per_block + 1):
                blk.resnets.append(ResnetBlock2D(prev if j == 0 else out_c, out_c, None, g, 1e-6))
            if i != len(rboc) - 1:
                blk.upsamplers = nn.ModuleList([Upsample2D(out_c)])
            self.up_blocks.append(blk)
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)


class _DownBlock(nn.Module):
    def __init__(self):
        super().__init__()
        self.resnets = nn.ModuleList()
        self.downsamplers = None


class EncoderDownsample2D(nn.Module):
    """``Downsample2D(use_conv=True, padding=0, name="op")``: zero row / column appended at the bottom / right, conv3x3 stride 2"""

    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=0)
        self._p = _Packed()

    def run(self, x: _Act):
        w = self._p.get("w", [self.conv.weight], lambda: pack_conv3x3(self.conv.weight.detach()))
        out = ops.conv3x3(x.t, w, x.b, x.h, x.w, x.c, stride=2, bias=self.conv.bias, pad_mode=1)
        return _Act(out, x.b, (x.h - 2) // 2 + 1, (x.w - 2) // 2 + 1, x.c)


class Encoder(nn.Module):
    """diffusers ``Encoder(double_z=True)``: conv_in, DownEncoderBlock2D x len(block_out_channels) (``layers_per_block``
    ResnetBlock2D(temb=None, eps 1e-6) + a padding-0 downsample on all but the last), UNetMidBlock2D (1 head), GroupNorm ->
    SiLU -> conv_out to 2 * latent_channels."""

    def __init__(self, cfg):
        super().__init__()
        boc = tuple(cfg.block_out_channels)
        g = cfg.norm_num_groups
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        prev = boc[0]
        for i, c in enumerate(boc):
            blk = _DownBlock()
            for j in range(cfg.layers_per_block):
                blk.resnets.append(ResnetBlock2D(prev if j == 0 else c, c, None, g, 1e-6))
            if i != len(boc) - 1:
                blk.downsamplers = nn.ModuleList([EncoderDownsample2D(c)])
            self.down_blocks.append(blk)
            prev = c
        mid = nn.Module()
        mid.resnets = nn.ModuleList([ResnetBlock2D(boc[-1], boc[-1], None, g, 1e-6), ResnetBlock2D(boc[-1], boc[-1], None, g, 1e-6)])
        mid.attentions = nn.ModuleList([VAEAttention(boc[-1], g, 1e-6)])
        self.mid_block = mid
        self.conv_norm_out = nn.GroupNorm(g, boc[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[-1], 2 * cfg.latent_channels, 3, padding=1)


class DiagonalGaussianDistribution:
    """diffusers 0.21.4 ``DiagonalGaussianDistribution`` over fp32 moments [B, 2C, h, w] on the device.  Like the
    reference (``generator = torch.manual_seed(...)``: a CPU generator, so ``randn_tensor`` draws on the host in the
    model dtype and moves the noise to the device) the random draw is host-side; the arithmetic is ``tg_gaussian_sample``."""

    def __init__(self, moments, dtype):
        self.parameters = moments
        self.dtype = dtype
        self._mean = self._logvar = None

    @property
    def mean(self):
        return self.parameters[:, : self.parameters.shape[1] // 2]

    @property
    def logvar(self):
        return torch.clamp(self.parameters[:, self.parameters.shape[1] // 2:], -30.0, 20.0)

    def sample(self, generator=None, scale=1.0):
        B, C2, h, w = self.parameters.shape
        if generator is not None and generator.device.type != "cpu":
            # a DEVICE generator (models/pipelines.py:594 `torch.Generator("cuda")`, stage 2): randn_tensor draws on the device in the model dtype
            noise = torch.randn((B, C2 // 2, h, w), generator=generator, device=self.parameters.device, dtype=self.dtype).to(torch.float32)
        else:
            noise = torch.randn((B, C2 // 2, h, w), generator=generator, dtype=self.dtype).to(device=self.parameters.device,
                                                                                               dtype=torch.float32)
        return ops.gaussian_sample(self.parameters, noise, scale).to(self.dtype)

    def mode(self, scale=1.0):
        return ops.gaussian_sample(self.parameters, None, scale).to(self.dtype)


class AutoencoderKLOutput(SimpleNamespace):
    pass


class DecoderOutput(SimpleNamespace):
    pass


class AutoencoderKL(nn.Module):
    def __init__(self, config=None, **kw):
        super().__init__()
        self.config = config if config is not None else sd_vae_config(**kw)
        lc = self.config.latent_channels
        self.encoder = Encoder(self.config)
        self.quant_conv = nn.Conv2d(2 * lc, 2 * lc, 1)
        self.post_quant_conv = nn.Conv2d(lc, lc, 1)
        self.decoder = Decoder(self.config)
        self._p = _Packed()
        for p_ in self.parameters():
            p_.requires_grad_(False)

    @property
    def dtype(self):
        return self.decoder.conv_in.weight.dtype

    @property
    def device(self):
        return self.decoder.conv_in.weight.device

    def encode(self, x, return_dict=True):
        """image NCHW in [-1, 1] -> ``AutoencoderKLOutput(latent_dist=DiagonalGaussianDistribution)`` (diffusers surface:
        ``vae.encode(image).latent_dist.sample(generator)``, reference models/pipelines.py:157, 624-625)"""
        if not x.is_cuda:
            raise RuntimeError("theatergen_amd VAE runs on the GPU only (no CPU fallback)")
        dt, enc, cfg = self.dtype, self.encoder, self.config
        B, _, H, W = x.shape
        w_in = self._p.get("e_conv_in", [enc.conv_in.weight], lambda: pack_conv3x3(enc.conv_in.weight.detach()))
        c = enc.conv_in.out_channels
        a = _Act(ops.conv_in(x.contiguous(), w_in, enc.conv_in.bias, c, dt), B, H, W, c)
        for blk in enc.down_blocks:
            for r in blk.resnets:
                a = r.run(a, None, None)
            if blk.downsamplers is not None:
                a = blk.downsamplers[0].run(a)
        a = enc.mid_block.resnets[0].run(a, None, None)
        a = enc.mid_block.attentions[0].run(a)
        a = enc.mid_block.resnets[1].run(a, None, None)
        y = ops.groupnorm(a.t, a.b, a.hw, cfg.norm_num_groups, 1e-6, enc.conv_norm_out.weight, enc.conv_norm_out.bias, silu=True)
        w_out = self._p.get("e_conv_out", [enc.conv_out.weight], lambda: pack_conv3x3(enc.conv_out.weight.detach()))
        h = ops.conv_out(y, w_out, enc.conv_out.bias, B, a.h, a.w, 2 * cfg.latent_channels, torch.float32)      # NCHW fp32
        qc = self._p.get("qc", [self.quant_conv.weight, self.quant_conv.bias],
                         lambda: (self.quant_conv.weight.detach().float().reshape(2 * cfg.latent_channels, 2 * cfg.latent_channels).contiguous(),
                                  self.quant_conv.bias.detach().float().contiguous()))
        moments = ops.conv1x1_nchw(h, qc[0], qc[1], 1.0)
        dist = DiagonalGaussianDistribution(moments, dt)
        if not return_dict:
            return (dist,)
        return AutoencoderKLOutput(latent_dist=dist)

    def encode_latents(self, image, generator=None):
        """the reference's ``encode`` helper (models/pipelines.py:131-160): ``scaling_factor * vae.encode(image).latent_dist.
        sample(generator)``, the factor folded into the sampling kernel"""
        return self.encode(image).latent_dist.sample(generator, scale=float(self.config.scaling_factor))

    def decode(self, z, return_dict=True, out_dtype=None):
        """z = latents ALREADY divided by ``config.scaling_factor`` (the reference's call convention) -> image NCHW"""
        return self._decode(z, 1.0, return_dict, out_dtype)

    def decode_latents(self, latents, return_dict=False, out_dtype=torch.float32):
        """loop latents -> image, the ``/ scaling_factor`` folded into the post-quant kernel"""
        return self._decode(latents, 1.0 / float(self.config.scaling_factor), return_dict, out_dtype)

    def _decode(self, z, in_scale, return_dict, out_dtype):
        if not z.is_cuda:
            raise RuntimeError("theatergen_amd VAE runs on the GPU only (no CPU fallback)")
        dt, dec, cfg = self.dtype, self.decoder, self.config
        B, _, H, W = z.shape
        pq = self._p.get("pq", [self.post_quant_conv.weight, self.post_quant_conv.bias],
                         lambda: (self.post_quant_conv.weight.detach().float().reshape(cfg.latent_channels, cfg.latent_channels).contiguous(),
                                  self.post_quant_conv.bias.detach().float().contiguous()))
        zq = ops.conv1x1_nchw(z.float().contiguous(), pq[0], pq[1], in_scale)
        w_in = self._p.get("conv_in", [dec.conv_in.weight], lambda: pack_conv3x3(dec.conv_in.weight.detach()))
        c = dec.conv_in.out_channels
        x = _Act(ops.conv_in(zq, w_in, dec.conv_in.bias, c, dt), B, H, W, c)
        x = dec.mid_block.resnets[0].run(x, None, None)
        x = dec.mid_block.attentions[0].run(x)
        x = dec.mid_block.resnets[1].run(x, None, None)
        for blk in dec.up_blocks:
            for r in blk.resnets:
                x = r.run(x, None, None)
            if blk.upsamplers is not None:
                x = blk.upsamplers[0].run(x)
        y = ops.groupnorm(x.t, x.b, x.hw, cfg.norm_num_groups, 1e-6, dec.conv_norm_out.weight, dec.conv_norm_out.bias, silu=True)
        w_out = self._p.get("conv_out", [dec.conv_out.weight], lambda: pack_conv3x3(dec.conv_out.weight.detach()))
        img = ops.conv_out(y, w_out, dec.conv_out.bias, B, x.h, x.w, cfg.out_channels, out_dtype if out_dtype is not None else dt)
        if not return_dict:
            return (img,)
        return DecoderOutput(sample=img)

    @classmethod
    def from_state_dict(cls, config, state_dict, device="cuda", dtype=torch.bfloat16):
        """a decoder-only state dict (no ``encoder.*`` keys) builds a decode-only module"""
        m = cls(config)
        if not any(k.startswith("encoder.") for k in state_dict):
            del m.encoder, m.quant_conv
            m.encoder = m.quant_conv = None
        missing, unexpected = m.load_state_dict(state_dict, strict=False)
        if unexpected or missing:
            raise RuntimeError(f"state dict mismatch: missing {missing[:5]}... unexpected {unexpected[:5]}...")
        return m.to(device=device, dtype=dtype)

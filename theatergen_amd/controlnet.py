"""``ControlNetModel`` on the native kernels — the stage-2 branch that runs every final-image step next to the UNet
(reference ``models/pipelines.py:698-731`` setup, ``:759-778`` per-step call, residuals consumed at
``models/unet_2d_condition.py:938-946, 975-976``).

Same call surface as ``diffusers.ControlNetModel`` (0.21.4) as the reference uses it:

    down_res, mid_res = controlnet(sample, t, encoder_hidden_states=text_embeds, controlnet_cond=image,
                                   conditioning_scale=s, guess_mode=False, return_dict=False)

The encoder half (conv_in, time embedding, down blocks, mid block) is built from the UNet's own modules
(``theatergen_amd/unet.py``) and therefore runs the same HIP kernels (LDS-halo conv, GEMM, flash attention, GroupNorm);
the 13 zero-convs are 1x1 GEMMs with ``conditioning_scale`` folded into the epilogue scale.  The conditioning
embedding (8 thin 3x3 convs on the 512x512 control image) is step-invariant: it is computed once per control image and
cached, on the MFMA conv kernel with channels zero-padded to multiples of 64 (the kernel's K granularity); SiLU rides
in the conv epilogue.  By default the residuals are handed to ``UNet2DConditionModel`` as token-major activations
(``token_major=True`` fast path, no NCHW round trip); ``token_major=False`` returns NCHW tensors like diffusers.
"""
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops
from .attention_processor import Attention, AttnProcessor, CNAttnProcessor, StaticSlots, tensor_version
from .config import UNetConfig
from .unet import (DeviceSchedule, Downsample2D, ResnetBlock2D, TimestepEmbedding, Transformer2DModel,
                   UNet2DConditionModel, _Act, _Block, _Packed)
from .weights_pack import pack_conv1x1, pack_conv3x3


def _pad64(c):
    return (c + 63) // 64 * 64


class ControlNetConditioningEmbedding(nn.Module):
    """diffusers 0.21.4 ``ControlNetConditioningEmbedding``: conv_in -> [conv(c, c), conv(c, c', stride 2)]* -> conv_out,
    SiLU after every conv but the last."""

    def __init__(self, conditioning_embedding_channels, conditioning_channels=3, block_out_channels=(16, 32, 96, 256)):
        super().__init__()
        boc = tuple(block_out_channels)
        self.conv_in = nn.Conv2d(conditioning_channels, boc[0], 3, padding=1)
        self.blocks = nn.ModuleList()
        for i in range(len(boc) - 1):
            self.blocks.append(nn.Conv2d(boc[i], boc[i], 3, padding=1))
            self.blocks.append(nn.Conv2d(boc[i], boc[i + 1], 3, padding=1, stride=2))
        self.conv_out = nn.Conv2d(boc[-1], conditioning_embedding_channels, 3, padding=1)
        nn.init.zeros_(self.conv_out.weight)          # zero_module(...) in diffusers
        nn.init.zeros_(self.conv_out.bias)
        self._p = _Packed()

    @staticmethod
    def _padded(conv, cin_p, cout_p):
        """zero-padded, tap-major packed weight + bias: padded output channels stay exactly 0 through SiLU"""
        w = conv.weight.detach()
        cout, cin = w.shape[:2]
        wp = torch.zeros((cout_p, cin_p, 3, 3), dtype=w.dtype, device=w.device)
        wp[:cout, :cin] = w
        bp = torch.zeros((cout_p,), dtype=w.dtype, device=w.device)
        bp[:cout] = conv.bias.detach()
        return pack_conv3x3(wp), bp

    def run(self, cond, dtype):
        """cond NCHW [B, 3, H, W] (any float dtype) -> token-major [B*(H/8)*(W/8), C_emb] activation"""
        B, _, H, W = cond.shape
        c0 = _pad64(self.conv_in.out_channels)
        w, b = self._p.get("in", [self.conv_in.weight, self.conv_in.bias], lambda: self._padded(self.conv_in, self.conv_in.in_channels, c0))
        x = ops.conv_in(cond.contiguous(), w, b, c0, dtype)
        x = ops.act(x, ops.ACT_SILU)
        h, wd, c = H, W, c0
        for i, conv in enumerate(self.blocks):
            cout_p = _pad64(conv.out_channels)
            wk, bk = self._p.get(f"b{i}", [conv.weight, conv.bias], lambda conv=conv, c=c, cout_p=cout_p: self._padded(conv, c, cout_p))
            stride = conv.stride[0]
            x = ops.conv3x3(x, wk, B, h, wd, c, stride=stride, bias=bk, act=ops.ACT_SILU)
            if stride == 2:
                h, wd = (h - 1) // 2 + 1, (wd - 1) // 2 + 1
            c = cout_p
        cout = self.conv_out.out_channels
        wo, bo = self._p.get("out", [self.conv_out.weight, self.conv_out.bias], lambda: self._padded(self.conv_out, c, cout))
        x = ops.conv3x3(x, wo, B, h, wd, c, bias=bo)
        return _Act(x, B, h, wd, cout)


class ControlNetOutput(SimpleNamespace):
    pass


class ControlNetModel(nn.Module):
    """SD encoder half + conditioning embedding + zero convs (diffusers 0.21.4 ``ControlNetModel`` semantics)."""

    def __init__(self, config: UNetConfig = None, conditioning_channels=3, conditioning_embedding_out_channels=(16, 32, 96, 256),
                 global_pool_conditions=False, **kw):
        super().__init__()
        cfg = config if config is not None else UNetConfig(**kw)
        if cfg.addition_embed_type is not None:
            raise NotImplementedError("ControlNet for the SDXL text_time plan is not on the SD-1.5 stage-2 path")
        self.config = cfg
        self.global_pool_conditions = bool(global_pool_conditions)
        boc = tuple(cfg.block_out_channels)
        nb = len(boc)
        ted = cfg.time_embed_dim
        heads_t = cfg.per_block(cfg.attention_head_dim)
        tl_t = cfg.per_block(cfg.transformer_layers_per_block)
        lpb = cfg.per_block(cfg.layers_per_block)
        g, eps, ctx, lin = cfg.norm_num_groups, cfg.norm_eps, cfg.cross_attention_dim, cfg.use_linear_projection

        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], ted)
        self.controlnet_cond_embedding = ControlNetConditioningEmbedding(boc[0], conditioning_channels, conditioning_embedding_out_channels)

        def tfm(c, heads, layers):
            return Transformer2DModel(heads, c // heads, c, layers, ctx, g, lin)

        def zero_conv(c):
            m = nn.Conv2d(c, c, 1)
            nn.init.zeros_(m.weight)
            nn.init.zeros_(m.bias)
            return m

        self.down_blocks = nn.ModuleList()
        self.controlnet_down_blocks = nn.ModuleList([zero_conv(boc[0])])
        out_c = boc[0]
        for i, bt in enumerate(cfg.down_block_types):
            in_c, out_c = out_c, boc[i]
            blk = _Block()
            if bt == "CrossAttnDownBlock2D":
                blk.has_cross_attention = True
                blk.attentions = nn.ModuleList()
            elif bt != "DownBlock2D":
                raise ValueError(f"unknown down block {bt}")
            for j in range(lpb[i]):
                blk.resnets.append(ResnetBlock2D(in_c if j == 0 else out_c, out_c, ted, g, eps))
                if blk.has_cross_attention:
                    blk.attentions.append(tfm(out_c, heads_t[i], tl_t[i]))
                self.controlnet_down_blocks.append(zero_conv(out_c))
            blk.downsamplers = nn.ModuleList([Downsample2D(out_c)]) if i != nb - 1 else None
            if blk.downsamplers is not None:
                self.controlnet_down_blocks.append(zero_conv(out_c))
            self.down_blocks.append(blk)

        mid = _Block()
        mid.has_cross_attention = True
        mid.resnets.append(ResnetBlock2D(boc[-1], boc[-1], ted, g, eps))
        mid.attentions = nn.ModuleList([tfm(boc[-1], heads_t[-1], tl_t[-1])])
        mid.resnets.append(ResnetBlock2D(boc[-1], boc[-1], ted, g, eps))
        self.mid_block = mid
        self.controlnet_mid_block = zero_conv(boc[-1])

        self._resnets = [m for m in self.modules() if isinstance(m, ResnetBlock2D)]
        off = 0
        for r in self._resnets:
            r.temb_slot = (off, r.out_channels)
            off += r.out_channels
        self._tproj_width = off
        self._p = _Packed()
        self._t_cache = {}
        self._cond_cache = StaticSlots()
        for p_ in self.parameters():
            p_.requires_grad_(False)

    # the time-embedding / fused time-projection plumbing and the processor table are the UNet's
    dtype = UNet2DConditionModel.dtype
    device = UNet2DConditionModel.device
    attn_processors = UNet2DConditionModel.attn_processors
    set_attn_processor = UNet2DConditionModel.set_attn_processor
    set_default_attn_processor = UNet2DConditionModel.set_default_attn_processor
    _timestep_dev = UNet2DConditionModel._timestep_dev
    _tproj_weight = UNet2DConditionModel._tproj_weight
    time_embed = UNet2DConditionModel.time_embed

    def cond_embedding(self, controlnet_cond, static=False):
        """Control-image embedding (step-invariant).  ``static=True`` is the owner of a persistent control-image buffer
        (``DenoiseEngine.set_control``) registering / refreshing it: the embedding then lives in a buffer that belongs to
        that tensor OBJECT (refreshed in place: a captured step graph keeps reading it; engines sharing this ControlNet
        each have their own).  Any other tensor is embedded on every call — identity is never inferred from an address."""
        emb = self.controlnet_cond_embedding
        wkey = (tuple(p._version for p in emb.parameters()), self.dtype)
        slot = self._cond_cache.get(controlnet_cond)
        if slot is not None and not static:
            ver = tensor_version(controlnet_cond)
            if ver is not None and slot["key"] == (ver, wkey):
                return slot["act"]
            static = True                                # registered tensor changed in place: refresh its buffer
        new = emb.run(controlnet_cond, self.dtype)
        if not static:
            return new
        if slot is not None and slot["act"].t.shape == new.t.shape and slot["act"].t.dtype == new.t.dtype:
            slot["act"].t.copy_(new.t)                    # IN PLACE: graphs captured on this slot stay valid
        else:
            slot = self._cond_cache.put(controlnet_cond, {"act": new})
        slot["key"] = (tensor_version(controlnet_cond), wkey)
        return slot["act"]

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale=1.0, class_labels=None,
                timestep_cond=None, attention_mask=None, cross_attention_kwargs=None, guess_mode=False, return_dict=True,
                token_major=False):
        cfg = self.config
        if attention_mask is not None or class_labels is not None or timestep_cond is not None:
            raise NotImplementedError("attention masks / class labels / timestep_cond are not on the TheaterGen stage-2 path")
        if not sample.is_cuda:
            raise RuntimeError("theatergen_amd ControlNet runs on the GPU only (no CPU fallback)")
        dt = self.dtype
        B, _, H, W = sample.shape
        enc = encoder_hidden_states
        if enc.dtype != dt:
            enc = enc.to(dt)
        enc = enc.contiguous()
        ca_kwargs = {} if cross_attention_kwargs is None else cross_attention_kwargs

        emb, tproj = self.time_embed(timestep, B, None)
        w_in = self._p.get("conv_in", [self.conv_in.weight], lambda: pack_conv3x3(self.conv_in.weight.detach()))
        c0 = cfg.block_out_channels[0]
        x = ops.conv_in(sample.contiguous(), w_in, self.conv_in.bias, c0, dt)
        ce = self.cond_embedding(controlnet_cond)
        if (ce.b, ce.h, ce.w, ce.c) != (B, H, W, c0):
            raise ValueError(f"controlnet_cond {tuple(controlnet_cond.shape)} does not map onto the latent grid {B}x{H}x{W} (x8)")
        x = _Act(ops.add(x, ce.t), B, H, W, c0)

        res = [x]
        for i, blk in enumerate(self.down_blocks):
            for j, resnet in enumerate(blk.resnets):
                x = resnet.run(x, None, tproj)
                if blk.has_cross_attention:
                    x = blk.attentions[j].run(x, enc, ca_kwargs)
                res.append(x)
            if blk.downsamplers is not None:
                x = blk.downsamplers[0].run(x)
                res.append(x)
        x = self.mid_block.resnets[0].run(x, None, tproj)
        x = self.mid_block.attentions[0].run(x, enc, ca_kwargs)
        x = self.mid_block.resnets[1].run(x, None, tproj)

        n_out = len(res) + 1
        if guess_mode and not self.global_pool_conditions:
            scales = [float(s) * float(conditioning_scale) for s in torch.logspace(-1, 0, n_out)]
        else:
            scales = [float(conditioning_scale)] * n_out

        def head(act, conv, name, scale):
            w = self._p.get(name, [conv.weight], lambda: pack_conv1x1(conv.weight.detach()))
            y = ops.gemm(act.t, w, act.b * act.hw, act.c, act.c, bias=conv.bias, out_scale=scale)
            return _Act(y, act.b, act.h, act.w, act.c)

        down = [head(r, self.controlnet_down_blocks[i], f"zc{i}", scales[i]) for i, r in enumerate(res)]
        mid = head(x, self.controlnet_mid_block, "zc_mid", scales[-1])
        if self.global_pool_conditions or not token_major:
            down = [self._nchw(a) for a in down]
            mid = self._nchw(mid)
            if self.global_pool_conditions:
                down = [d.float().mean(dim=(2, 3), keepdim=True).to(dt) for d in down]
                mid = mid.float().mean(dim=(2, 3), keepdim=True).to(dt)
        if not return_dict:
            return (down, mid)
        return ControlNetOutput(down_block_res_samples=down, mid_block_res_sample=mid)

    __call__ = forward

    @staticmethod
    def _nchw(a: _Act):
        return ops.transpose(a.t.reshape(a.b, a.hw, a.c), a.b, a.hw, a.c).reshape(a.b, a.c, a.h, a.w)

    @classmethod
    def from_unet(cls, unet, **kw):
        """diffusers' ``ControlNetModel.from_unet``: same plan, encoder weights copied from the UNet."""
        m = cls(unet.config, **kw)
        sd = {k: v for k, v in unet.state_dict().items()
              if k.startswith(("conv_in.", "time_embedding.", "down_blocks.", "mid_block.")) and ".processor." not in k}
        missing, unexpected = m.load_state_dict(sd, strict=False)
        if unexpected:
            raise RuntimeError(f"from_unet: unexpected keys {unexpected[:5]}")
        return m.to(device=unet.device, dtype=unet.dtype)

    @classmethod
    def from_state_dict(cls, config, state_dict, device="cuda", dtype=torch.bfloat16, cn_processors=True, num_tokens=4, **kw):
        from .unet import _load_on_meta
        m = _load_on_meta(lambda: cls(config, **kw), state_dict, device, dtype)
        if cn_processors:
            install_cn_processors(m, num_tokens)
        return m


def install_cn_processors(controlnet, num_tokens=4):
    """``IPAdapter.set_ip_adapter`` puts ``CNAttnProcessor`` on the ControlNet's cross-attention so the image tokens
    appended to ``encoder_hidden_states`` are ignored there (reference ip_adapter/ip_adapter.py:115-119)."""
    procs = {}
    for name in controlnet.attn_processors.keys():
        procs[name] = AttnProcessor() if name.endswith("attn1.processor") else CNAttnProcessor(num_tokens=num_tokens)
    controlnet.set_attn_processor(procs)
    return procs

"""MI355X-native ``UNet2DConditionModel`` with the diffusers call surface the reference uses.

Structure (block order, channel plan, where attention sits) follows the reference's UNet specification:
``models/unet_2d_condition.py`` (ctor :209-593, ``forward`` :725-1023), ``models/unet_2d_blocks.py``
(CrossAttnDownBlock2D :279-451, DownBlock2D :454-537, UNetMidBlock2DCrossAttn :155-276, CrossAttnUpBlock2D
:540-713, UpBlock2D :716-797), ``models/transformer_2d.py`` (:216-370), ``models/attention.py``
(BasicTransformerBlock :156-240, FeedForward/GEGLU :243-338); SDXL ``text_time`` embedding
``ip_adapter/unet_2d_condition.py:937-954``.  Module / parameter names equal the diffusers state-dict keys, so
``load_state_dict`` of an SD-1.5 / SD-2.1 / SDXL checkpoint works, and ``unet.attn_processors`` /
``unet.set_attn_processor`` behave as in the reference (``models/unet_2d_condition.py:596-648``).

Execution is MI355X-first, not a module-by-module translation:
  * activations stay token-major ("NHWC") [B*h*w, C] end to end: the NCHW<->[B,HW,C] permutes of
    Transformer2DModel (transformer_2d.py:289, 324) and the skip-connection ``torch.cat`` (unet_2d_blocks.py:648-651)
    do not exist — GroupNorm and the implicit-GEMM conv read two sources;
  * every conv / linear is the MFMA GEMM (``tg_gemm``) with bias, time-embedding add, residual add and scale in
    its epilogue; the 22 ResBlock time projections are ONE GEMM per step;
  * attention is the fused flash kernel (no [B*h, N, N] tensor); text / image K, V^T are projected once per
    conditioning, not once per step;
  * ``torch.nn`` modules are parameter containers only.  No PyTorch compute, no CPU fallback.
"""
from types import SimpleNamespace

import os

import torch
import torch.nn as nn

from . import ops
from . import rowchain
from .attention_processor import (tensor_version, Attention, AttnProcessor, CNAttnProcessor, IPAttnProcessor, _cached, front_eligible, fused_cross_block,
                                  ln_weight, self_attention_from_qkv)
from .config import UNetConfig
from .weights_pack import pack_conv1x1, pack_conv3x3, pack_geglu, pack_ln_linear, rc_pack_tiles


class DeviceSchedule:
    """Device-resident timestep table + step counter: pass as ``timestep`` so a captured step graph replays
    without host-side updates (``table`` fp32 [n_steps], ``index`` int32 [1] advanced by ``tg_step_epilogue``)."""

    def __init__(self, table, index):
        self.table, self.index = table, index


class _Act:
    """A token-major activation: 2-D tensor [B*h*w, C] + its geometry."""
    __slots__ = ("t", "b", "h", "w", "c", "gn")

    def __init__(self, t, b, h, w, c, gn=None):
        self.t, self.b, self.h, self.w, self.c = t, b, h, w, c
        self.gn = gn          # GroupNorm partial sums of ``t`` its producer conv wrote (ops.gemm gn_out), or None

    @property
    def hw(self):
        return self.h * self.w


class _Packed:
    """Kernel-layout copies of a module's weights, rebuilt when the parameters change (load_state_dict / .to)."""

    def __init__(self):
        self._c = {}

    def get(self, name, tensors, build):
        key = tuple((t.data_ptr(), tensor_version(t), t.dtype, t.device) for t in tensors)
        hit = self._c.get(name)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = (key, build())
            self._c[name] = hit
        return hit[1]


# dev switch: TG_NO_GN_FUSE=1 keeps GroupNorm and conv two ops everywhere (A/B of the fused window staging)
_FUSE_GN = not os.environ.get("TG_NO_GN_FUSE")
# round 6: the statistics pass of a GroupNorm whose input is a slab conv's output (ResnetBlock2D.norm2 after conv1, Transformer2DModel.norm after conv2) comes
# out of that conv's epilogue as partial sums (tg_gemm_desc.out_gn_partials) where the two-wave slab kernel runs the layer unsplit.  TG_GN_EPI=0 (dev A/B) = off.
_GN_EPI = os.environ.get("TG_GN_EPI", "1") != "0"


def _gn_of(gn, groups):
    return gn if gn is not None and "partials" in gn and gn["groups"] == groups else None
# dev switch: TG_NO_LN_FUSE=1 keeps LayerNorm a launch of its own in front of the q|k|v / to_q / GEGLU projections
# TG_LN_MODE bits: 1 = fold norm1 / norm2 into attn1 q|k|v / attn2.to_q, 2 = fold norm3 into the GEGLU GEMM, 4 = row statistics from a
# statistics-only pass (tg_layernorm_stats) instead of inside the GEMM.  0 = LayerNorm launches (round 2).
# Default 1, by same-box graph-replay A/B (profiles/r3_ln_findings.md): attention-only fold with in-kernel statistics +1.06 % images/s;
# folding norm3 too costs the FeedForward its 256 x 256 big-tile GEGLU kernel (-0.4 % overall); a separate statistics pass is slower
# than taking them from the staged tiles (+0.75 % instead of +1.06 %).
_LN_MODE = 0 if os.environ.get("TG_NO_LN_FUSE") else int(os.environ.get("TG_LN_MODE", "1"))
# norm3 folded into the GEGLU GEMM for row counts up to this (0 = never): where the plain GEGLU GEMM runs on the 128 x 128 kernel anyway (not the 256 x 256 big tile of
# the 32 x 32 level) the fold costs no tile choice and removes the layernorm launch
_LN_FF_MAX_ROWS = int(os.environ.get("TG_LN_FF_MAX_ROWS", "0"))
_CONV_OUT_GN = os.environ.get("TG_CONV_OUT_GN", "1") != "0"      # dev A/B knob: 0 = GroupNorm apply launch + conv_out
_FF_PAD = os.environ.get("TG_FF_PAD", "1") != "0"      # round 5: pad the FeedForward hidden tensor's / net.2 weight's row pitch (see FeedForward._hidden)
_PP_MODE = int(os.environ.get("TG_PP", "15"))           # csrc/tg_gemm.hip pp_mode(): bit 2 = LayerNorm-folded projections on the ping-pong tiles


def _pp_takes_ln(M, N, K):
    """mirror of csrc/tg_gemm.hip ``pp_selected`` for a LayerNorm-folded projection (the kernel needs the row statistics as an input)"""
    if not (_PP_MODE & 4) or M % 256 or K % 64:
        return False
    if N % 256 == 0 and K >= 640:                                      # 256 x 256 tiles (pp_selected)
        tiles = (M // 256) * (N // 256)
        if tiles >= 192 and tiles / (((tiles + 255) // 256) * 256) >= 0.6:
            return True
    if N % 160 == 0 and K >= 640 and (_PP_MODE & 8):                   # 256 x 160 tiles (pp160_selected); q | k | v^T: n_split = 2 N / 3 on an 80-column boundary
        tiles = (M // 256) * (N // 160)
        return tiles >= 192 and tiles / (((tiles + 255) // 256) * 256) >= 0.74 and (2 * N // 3) % 80 == 0
    return False


_FUSE_LN_MIN_ROWS = int(os.environ.get("TG_LN_FUSE_MIN_ROWS", "2048"))     # below: few 128-row tiles, the 64 x 64-tile path wins


class ResnetBlock2D(nn.Module):
    """diffusers 0.21.4 ResnetBlock2D semantics: h = conv1(silu(GN(x))) + Linear(silu(temb));
    h = conv2(silu(GN(h))); out = (shortcut(x) + h) / output_scale_factor."""

    def __init__(self, in_channels, out_channels, temb_channels, groups=32, eps=1e-5, output_scale_factor=1.0):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.groups, self.eps, self.output_scale_factor = groups, eps, output_scale_factor
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        # temb_channels None: the VAE's ResnetBlock2D(temb_channels=None) has no time projection
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None
        self._p = _Packed()
        self.temb_slot = None      # (offset, width) into the fused time-projection output, set by the UNet

    def run(self, x: _Act, skip: _Act, tproj):
        """x (+ skip = channel concat) -> _Act.  ``tproj`` = fused time projections [B, sum(cout)]."""
        c1 = skip.c if skip is not None else 0
        assert x.c + c1 == self.in_channels
        w1 = self._p.get("c1", [self.conv1.weight], lambda: pack_conv3x3(self.conv1.weight.detach()))
        w2 = self._p.get("c2", [self.conv2.weight], lambda: pack_conv3x3(self.conv2.weight.detach()))
        # GroupNorm + SiLU ride in the conv's window staging where the slab conv kernel takes the layer (N % 320 == 0, enough
        # tiles to fill the chip): only the statistics pass (-> per-(image, channel) a, d) remains a launch of its own and
        # the normalised tensor never reaches HBM; elsewhere norm and conv stay two ops.  Same values bit for bit.
        x1 = skip.t if skip is not None else None
        kw1 = dict(bias=self.conv1.bias)
        if tproj is not None:
            off, width = self.temb_slot
            kw1.update(bvec=tproj[:, off:off + width], rows_per_batch=x.hw)
        fuse2 = _FUSE_GN and ops.conv3x3_takes_gn(x.t.dtype, x.b, x.h, x.w, self.out_channels, 0, self.out_channels)
        gn1 = {"groups": self.groups} if _GN_EPI and fuse2 else None      # conv1 -> norm2: the coefficients are all norm2 needs
        if gn1 is not None:
            kw1.update(gn_out=gn1)
        if _FUSE_GN and ops.conv3x3_takes_gn(x.t.dtype, x.b, x.h, x.w, x.c, c1, self.out_channels):
            g0 = _gn_of(x.gn, self.groups) if skip is None else None
            coef = (ops.groupnorm_from_partials(g0, x.b, x.hw, x.c, self.eps, self.norm1.weight, self.norm1.bias) if g0 is not None else
                    ops.groupnorm_coef(x.t, x.b, x.hw, self.groups, self.eps, self.norm1.weight, self.norm1.bias, x1=x1))
            h = ops.conv3x3(x.t, w1, x.b, x.h, x.w, x.c, x1=x1, c1=c1, a_coef=coef, a_silu=True, **kw1)
        else:
            h = ops.groupnorm(x.t, x.b, x.hw, self.groups, self.eps, self.norm1.weight, self.norm1.bias, silu=True, x1=x1)
            h = ops.conv3x3(h, w1, x.b, x.h, x.w, self.in_channels, **kw1)
        if self.conv_shortcut is not None:
            ws = self._p.get("sc", [self.conv_shortcut.weight], lambda: pack_conv1x1(self.conv_shortcut.weight.detach()))
            res = ops.gemm(x.t, ws, x.b * x.hw, self.out_channels, self.in_channels, a1=skip.t if skip is not None else None,
                           c0=x.c, c1=c1, bias=self.conv_shortcut.bias)
        else:
            res = x.t
        kw2 = dict(bias=self.conv2.bias, res=res, out_scale=1.0 / self.output_scale_factor)
        gn2 = {"groups": self.groups} if _GN_EPI else None                 # conv2 -> the next GroupNorm(groups) of the block output (Transformer2DModel.norm)
        if gn2 is not None:
            kw2.update(gn_out=gn2)
        if fuse2:
            g1 = _gn_of(gn1, self.groups)
            coef = (ops.groupnorm_from_partials(g1, x.b, x.hw, self.out_channels, self.eps, self.norm2.weight, self.norm2.bias) if g1 is not None else
                    ops.groupnorm_coef(h, x.b, x.hw, self.groups, self.eps, self.norm2.weight, self.norm2.bias))
            out = ops.conv3x3(h, w2, x.b, x.h, x.w, self.out_channels, a_coef=coef, a_silu=True, **kw2)
        else:
            h = ops.groupnorm(h, x.b, x.hw, self.groups, self.eps, self.norm2.weight, self.norm2.bias, silu=True)
            out = ops.conv3x3(h, w2, x.b, x.h, x.w, self.out_channels, **kw2)
        return _Act(out, x.b, x.h, x.w, self.out_channels, gn=_gn_of(gn2, self.groups))


class Downsample2D(nn.Module):
    """conv3x3 stride 2, padding 1 (``Downsample2D(use_conv=True, padding=1, name="op")``, unet_2d_blocks.py:355-362)."""

    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)
        self._p = _Packed()

    def run(self, x: _Act):
        w = self._p.get("w", [self.conv.weight], lambda: pack_conv3x3(self.conv.weight.detach()))
        out = ops.conv3x3(x.t, w, x.b, x.h, x.w, x.c, stride=2, bias=self.conv.bias)
        return _Act(out, x.b, (x.h - 1) // 2 + 1, (x.w - 1) // 2 + 1, x.c)


class Upsample2D(nn.Module):
    """nearest x2 folded into the conv3x3 gather (``Upsample2D(use_conv=True)``, unet_2d_blocks.py:620-622)."""

    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)
        self._p = _Packed()

    def run(self, x: _Act):
        w = self._p.get("w", [self.conv.weight], lambda: pack_conv3x3(self.conv.weight.detach()))
        out = ops.conv3x3(x.t, w, x.b, x.h, x.w, x.c, upsample=True, bias=self.conv.bias)
        return _Act(out, x.b, 2 * x.h, 2 * x.w, x.c)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    """net.0 = GEGLU(C -> 4C), net.1 = Dropout, net.2 = Linear(4C -> C)  (models/attention.py:243-292)."""

    def __init__(self, dim, mult=4):
        super().__init__()
        inner = dim * mult
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(0.0), nn.Linear(inner, dim)])
        self._p = _Packed()

    def run_fused(self, x2d, norm, tail=None):
        """norm3 + GEGLU feed-forward + residual — and with ``tail`` = (packed proj_out weight [C, C], bias, its residual) also
        ``Transformer2DModel.proj_out`` + residual — as ONE row-chain launch (``tg_rc_ff``; first level of SD-1.5: 320 channels).
        ``x2d`` is the un-normalised stream.  Returns None when the shape is not eligible."""
        proj, lin2 = self.net[0].proj, self.net[2]
        M, K = x2d.shape
        inner = lin2.weight.shape[1]
        if (not rowchain.ENABLED or not (rowchain.MODE & 4) or K != 320 or lin2.weight.shape[0] != 320 or inner % 64 or M < rowchain.MIN_ROWS_CHAIN
                or x2d.stride(0) != K or proj.weight.shape[0] != 2 * inner):
            return None
        ts = [proj.weight, proj.bias, norm.weight, norm.bias, lin2.weight] + ([lin2.bias] if lin2.bias is not None else [])
        s1, s2, b2 = self._p.get("rc_ff", ts, lambda: rowchain.pack_ff(proj.weight, proj.bias, norm.weight, norm.bias, lin2.weight, lin2.bias))
        wpo = res0 = None
        if tail is not None:
            w_out, b_out, res0 = tail
            wpo = self._p.get("rc_po", [w_out] + ([b_out] if b_out is not None else []),
                              lambda: rc_pack_tiles(w_out.detach(), b_out.detach().float() if b_out is not None else None))
        return ops.rc_ff(x2d, s1, s2, b2, inner, norm.eps, wpo=wpo, res0=res0)

    def _hidden(self, M, inner, like):
        """the [M, inner] hidden tensor.  Round 5: when its row pitch would be a multiple of 1 KiB (inner = 2560 / 5120: the 32 x 32 / 16 x 16 levels) the rows are
        padded by 64 elements (and net.2's weight likewise, ``_w2``) — its only consumers are the two FeedForward GEMMs, and the lockstep workgroups of those
        single-round launches otherwise crowd a few L2 channels (profiles/r5_operand_pitch.txt); TG_FF_PAD=0 switches it off"""
        if _FF_PAD and inner % 512 == 0:
            return torch.empty((M, inner + 64), dtype=like.dtype, device=like.device)[:, :inner]
        return torch.empty((M, inner), dtype=like.dtype, device=like.device)

    def _w2(self):
        w2 = self.net[2].weight
        inner = w2.shape[1]
        if not (_FF_PAD and inner % 512 == 0):
            return w2

        def build():
            wp = torch.zeros((w2.shape[0], inner + 64), dtype=w2.dtype, device=w2.device)
            wp[:, :inner] = w2.detach()
            return wp
        return self._p.get("w2pad", [w2], build)[:, :inner]

    def run(self, x2d, residual, ln=None):
        """``ln`` = (nn.LayerNorm, row statistics or None): ``x2d`` is the un-normalised stream and the norm is folded into the GEGLU GEMM"""
        proj = self.net[0].proj
        M, K = x2d.shape
        inner = proj.weight.shape[0] // 2
        if ln is not None:
            norm, rows = ln

            def build():
                wp, bp = pack_geglu(proj.weight.detach(), proj.bias.detach())
                return pack_ln_linear(wp, bp, norm.weight, norm.bias)
            wl, ul, vl = self._p.get("geglu_ln", [proj.weight, proj.bias, norm.weight, norm.bias], build)
            g = ops.gemm(x2d, wl, M, wl.shape[0], K, geglu=True, ln=(ul, vl, norm.eps, rows), out=self._hidden(M, inner, x2d))
        elif M > 64 and inner % 32 == 0:
            # GEGLU fused into the C -> 8C GEMM epilogue: the [M, 8C] pre-activation is never written
            wp, bp = self._p.get("geglu", [proj.weight, proj.bias], lambda: pack_geglu(proj.weight.detach(), proj.bias.detach()))
            g = ops.gemm(x2d, wp, M, wp.shape[0], K, bias=bp, geglu=True, out=self._hidden(M, inner, x2d))
        else:
            g = ops.geglu(ops.linear(x2d, proj.weight, proj.bias))
            return ops.linear(g, self.net[2].weight, self.net[2].bias, res=residual)
        return ops.linear(g, self._w2(), self.net[2].bias, res=residual)


class BasicTransformerBlock(nn.Module):
    """LN -> attn1 (self) ; LN -> attn2 (cross) ; LN -> GEGLU FF, all residual (models/attention.py:156-240)."""

    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(query_dim=dim, heads=heads, dim_head=dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(query_dim=dim, cross_attention_dim=cross_attention_dim, heads=heads, dim_head=dim_head)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    @staticmethod
    def _call(attn, x2d, b, n, enc, residual, kwargs, ln=None):
        """``ln`` None: ``x2d`` is the normalised input.  ``ln`` = (the block's ``nn.LayerNorm``, row statistics or None): ``x2d`` is the
        un-normalised stream; our own processors fold the norm into their first projection, any other processor gets ``tg_layernorm`` first."""
        proc = attn.processor
        ours = isinstance(proc, (AttnProcessor, IPAttnProcessor, CNAttnProcessor))
        if ln is not None and not ours:
            x2d, ln = ops.layernorm(x2d, ln[0].weight, ln[0].bias, ln[0].eps), None
        x3 = x2d.reshape(b, n, -1)
        if isinstance(proc, (AttnProcessor, IPAttnProcessor)) and attn.rescale_output_factor == 1.0 \
                and not attn.residual_connection and not kwargs.get("return_attntion_probs"):
            return proc(attn, x3, encoder_hidden_states=enc, _fused_residual=residual, _fused_ln=ln, **kwargs).reshape(b * n, -1)
        if isinstance(proc, CNAttnProcessor):
            return proc(attn, x3, encoder_hidden_states=enc, _fused_residual=residual, _fused_ln=ln).reshape(b * n, -1)
        if ln is not None:
            x3 = ops.layernorm(x2d, ln[0].weight, ln[0].bias, ln[0].eps).reshape(b, n, -1)
        # foreign processor: plain diffusers protocol, residual added afterwards
        out = attn(x3, encoder_hidden_states=enc, **kwargs)
        if isinstance(out, tuple):
            out = out[0]
        return ops.add(out.reshape(b * n, -1).contiguous(), residual)

    def run(self, x2d, b, n, enc, ca_kwargs, tail=None, pre_qkv=None):
        """``tail`` (optional, from ``Transformer2DModel``: (packed proj_out weight, bias, residual)) — when the fused feed-forward launch
        takes it, the block returns ``(proj_out output, True)``; otherwise a plain tensor (the caller runs proj_out itself).
        ``pre_qkv`` = (Q | K, V^T, ldt) of norm1(x2d) already projected by ``tg_rc_front``."""
        M, C = x2d.shape
        if _LN_MODE and M >= _FUSE_LN_MIN_ROWS and C % 64 == 0 and x2d.stride(0) == C:
            # LayerNorm rides in the projection that consumes it (tg_gemm ln_u / ln_v): no normalised tensor.  Row statistics: a
            # statistics-only pass (half the layernorm kernel's traffic) or taken inside the GEMM from its own A tiles
            def folded(norm, t, n_out=0):
                # round 6: where the ping-pong 256 x 256 GEMM takes the projection (tg_gemm_pp.hip: it folds the LayerNorm from PRECOMPUTED row statistics)
                # the statistics-only pass runs first: 16 x 16 level q | k | v, 4096 x 3840 x 1280: 75 -> 54 + 6 us
                return (norm, ops.layernorm_stats(t, norm.eps) if (_LN_MODE & 4 or _pp_takes_ln(t.shape[0], n_out, t.shape[1])) else None)
            if _LN_MODE & 1:
                if pre_qkv is not None:
                    x2d = self_attention_from_qkv(self.attn1, pre_qkv[0], pre_qkv[1], pre_qkv[2], b, n, x2d)
                else:
                    x2d = self._call(self.attn1, x2d, b, n, None, x2d, ca_kwargs, ln=folded(self.norm1, x2d, 3 * self.attn1.to_q.weight.shape[0]))
                # first level of SD-1.5: norm2 + attn2 + residual as ONE row-chain launch (q, scores and O stay in registers)
                h2 = fused_cross_block(self.attn2, self.norm2, x2d, b, n, enc, ca_kwargs)
                x2d = h2 if h2 is not None else self._call(self.attn2, x2d, b, n, enc, x2d, ca_kwargs, ln=folded(self.norm2, x2d))
            else:
                h = ops.layernorm(x2d, self.norm1.weight, self.norm1.bias, self.norm1.eps)
                x2d = self._call(self.attn1, h, b, n, None, x2d, ca_kwargs)
                h = ops.layernorm(x2d, self.norm2.weight, self.norm2.bias, self.norm2.eps)
                x2d = self._call(self.attn2, h, b, n, enc, x2d, ca_kwargs)
            fused = self.ff.run_fused(x2d, self.norm3, tail)
            if fused is not None:
                return (fused, True) if tail is not None else fused
            if _LN_MODE & 2 or M <= _LN_FF_MAX_ROWS:
                return self.ff.run(x2d, x2d, ln=folded(self.norm3, x2d))
            h = ops.layernorm(x2d, self.norm3.weight, self.norm3.bias, self.norm3.eps)
            return self.ff.run(h, x2d)
        h = ops.layernorm(x2d, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        x2d = self._call(self.attn1, h, b, n, None, x2d, ca_kwargs)
        h = ops.layernorm(x2d, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        x2d = self._call(self.attn2, h, b, n, enc, x2d, ca_kwargs)
        h = ops.layernorm(x2d, self.norm3.weight, self.norm3.bias, self.norm3.eps)
        return self.ff.run(h, x2d)


class Transformer2DModel(nn.Module):
    """GroupNorm(eps 1e-6) -> proj_in -> N x BasicTransformerBlock -> proj_out -> + residual (transformer_2d.py:216-370)."""

    def __init__(self, heads, dim_head, in_channels, num_layers, cross_attention_dim, groups, use_linear_projection):
        super().__init__()
        inner = heads * dim_head
        assert inner == in_channels
        self.groups = groups
        self.use_linear_projection = use_linear_projection
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        if use_linear_projection:
            self.proj_in = nn.Linear(in_channels, inner)
            self.proj_out = nn.Linear(inner, in_channels)
        else:
            self.proj_in = nn.Conv2d(in_channels, inner, 1)
            self.proj_out = nn.Conv2d(inner, in_channels, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim) for _ in range(num_layers)])
        self._p = _Packed()

    def _w(self, name, mod):
        if self.use_linear_projection:
            return mod.weight
        return self._p.get(name, [mod.weight], lambda: pack_conv1x1(mod.weight.detach()))

    def _front(self, x, w_in, ca_kwargs):
        """GroupNorm (statistics launch + coefficients) + proj_in + the first block's norm1 + q | k | v as ONE row-chain launch
        (``tg_rc_front``; first level of SD-1.5) -> (y, Q | K, V^T, ldt), or None when the shapes / processors are not eligible."""
        blk = self.transformer_blocks[0]
        M, C = x.t.shape
        if (not rowchain.ENABLED or not (rowchain.MODE & 8) or not _LN_MODE & 1 or C != 320 or w_in.shape != (320, 320) or M < rowchain.MIN_ROWS_CHAIN
                or x.hw % 128 or x.t.stride(0) != C or x.c != 320 or not front_eligible(blk.attn1, ca_kwargs)):
            return None
        g = _gn_of(x.gn, self.groups)
        coef = (ops.groupnorm_from_partials(g, x.b, x.hw, x.c, 1e-6, self.norm.weight, self.norm.bias) if g is not None else
                ops.groupnorm_coef(x.t, x.b, x.hw, self.groups, 1e-6, self.norm.weight, self.norm.bias))
        bin_ = self.proj_in.bias
        win = self._p.get("rc_front_in", [w_in] + ([bin_] if bin_ is not None else []),
                          lambda: rc_pack_tiles(w_in.detach(), bin_.detach().float() if bin_ is not None else None))
        wl, ul, vl = ln_weight(blk.attn1, "qkv", blk.norm1)
        wqkv = self._p.get("rc_front_qkv", [wl, ul, vl], lambda: rc_pack_tiles(wl, vl, ul))
        return ops.rc_front(x.t, coef, win, wqkv, x.hw, blk.norm1.eps)

    def run(self, x: _Act, enc, ca_kwargs):
        w_in = self._w("in", self.proj_in)
        pre = self._front(x, w_in, ca_kwargs)
        if pre is not None:
            y, pre = pre[0], pre[1:]
        else:
            g = _gn_of(x.gn, self.groups)
            y = (ops.groupnorm_from_partials(g, x.b, x.hw, x.c, 1e-6, self.norm.weight, self.norm.bias, x=x.t) if g is not None else
                 ops.groupnorm(x.t, x.b, x.hw, self.groups, 1e-6, self.norm.weight, self.norm.bias, silu=False))
            y_in = rowchain.linear320(y, w_in, self.proj_in.bias, None, self, "proj_in", _cached)
            y = y_in if y_in is not None else ops.linear(y, w_in, self.proj_in.bias)
        base_key = list(ca_kwargs.get("attn_key", [])) if "attn_key" in ca_kwargs else None
        w_out = self._w("out", self.proj_out)
        last = len(self.transformer_blocks) - 1
        for i, blk in enumerate(self.transformer_blocks):
            if base_key is not None:
                ca_kwargs["attn_key"] = base_key + [i]           # transformer_2d.py:299-304
            y = blk.run(y, x.b, x.hw, enc, ca_kwargs, tail=(w_out, self.proj_out.bias, x.t) if i == last else None,
                        pre_qkv=pre if i == 0 else None)
            if isinstance(y, tuple):                             # the last block's fused feed-forward launch ran proj_out + residual too
                return _Act(y[0], x.b, x.h, x.w, x.c)
        out = rowchain.linear320(y, w_out, self.proj_out.bias, x.t, self, "proj_out", _cached)
        if out is None:
            out = ops.linear(y, w_out, self.proj_out.bias, res=x.t)
        return _Act(out, x.b, x.h, x.w, x.c)


class _Block(nn.Module):
    def __init__(self):
        super().__init__()
        self.resnets = nn.ModuleList()
        self.has_cross_attention = False


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim, dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def run(self, x2d):
        h = ops.linear(x2d, self.linear_1.weight, self.linear_1.bias, act=ops.ACT_SILU)
        return ops.linear(h, self.linear_2.weight, self.linear_2.bias)


class UNet2DConditionOutput(SimpleNamespace):
    pass


class UNet2DConditionModel(nn.Module):
    def __init__(self, config: UNetConfig = None, **kw):
        super().__init__()
        cfg = config if config is not None else UNetConfig(**kw)
        self.config = cfg
        boc = tuple(cfg.block_out_channels)
        nb = len(boc)
        ted = cfg.time_embed_dim
        heads_t = cfg.per_block(cfg.attention_head_dim)
        tl_t = cfg.per_block(cfg.transformer_layers_per_block)
        lpb = cfg.per_block(cfg.layers_per_block)
        g, eps, ctx, lin = cfg.norm_num_groups, cfg.norm_eps, cfg.cross_attention_dim, cfg.use_linear_projection

        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], ted)
        if cfg.addition_embed_type == "text_time":
            self.add_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim, ted)
        elif cfg.addition_embed_type is not None:
            raise ValueError(f"addition_embed_type {cfg.addition_embed_type} unsupported")

        def tfm(c, heads, layers):
            return Transformer2DModel(heads, c // heads, c, layers, ctx, g, lin)

        # registration order = the reference's (models/unet_2d_condition.py:430-431: both lists exist before mid_block), so
        # ``attn_processors`` enumerates down (0-11), up (12-29), mid (30-31) for SD-1.5 and the IP-Adapter checkpoints'
        # ``ip_adapter.{1,3,...,31}.to_k_ip.weight`` keys (ip_adapter.py:139-140) land on the layers they were trained for
        self.down_blocks = nn.ModuleList()
        self.up_blocks = nn.ModuleList()
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
            blk.downsamplers = nn.ModuleList([Downsample2D(out_c)]) if i != nb - 1 else None
            self.down_blocks.append(blk)

        mid = _Block()
        mid.has_cross_attention = True
        mid.resnets.append(ResnetBlock2D(boc[-1], boc[-1], ted, g, eps))
        mid.attentions = nn.ModuleList([tfm(boc[-1], heads_t[-1], tl_t[-1])])
        mid.resnets.append(ResnetBlock2D(boc[-1], boc[-1], ted, g, eps))
        self.mid_block = mid

        rboc, rheads, rtl, rlpb = tuple(reversed(boc)), tuple(reversed(heads_t)), tuple(reversed(tl_t)), tuple(reversed(lpb))
        out_c = rboc[0]
        for i, bt in enumerate(cfg.up_block_types):
            prev_out, out_c = out_c, rboc[i]
            in_c = rboc[min(i + 1, nb - 1)]
            blk = _Block()
            if bt == "CrossAttnUpBlock2D":
                blk.has_cross_attention = True
                blk.attentions = nn.ModuleList()
            elif bt != "UpBlock2D":
                raise ValueError(f"unknown up block {bt}")
            n = rlpb[i] + 1
            for j in range(n):
                skip_c = in_c if j == n - 1 else out_c
                res_in = prev_out if j == 0 else out_c
                blk.resnets.append(ResnetBlock2D(res_in + skip_c, out_c, ted, g, eps))
                if blk.has_cross_attention:
                    blk.attentions.append(tfm(out_c, rheads[i], rtl[i]))
            blk.upsamplers = nn.ModuleList([Upsample2D(out_c)]) if i != nb - 1 else None
            self.up_blocks.append(blk)

        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=eps)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

        # fused time projections: one GEMM for all ResBlocks
        self._resnets = [m for m in self.modules() if isinstance(m, ResnetBlock2D)]
        off = 0
        for r in self._resnets:
            r.temb_slot = (off, r.out_channels)
            off += r.out_channels
        self._tproj_width = off
        self._p = _Packed()
        self._t_cache = {}
        for p_ in self.parameters():
            p_.requires_grad_(False)

    # ---- diffusers surface ----------------------------------------------------------------------------
    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    @property
    def attn_processors(self):
        """{"...attn1.processor": proc, ...} in module order (models/unet_2d_condition.py:596-618)."""
        procs = {}
        for name, m in self.named_modules():
            if isinstance(m, Attention):
                procs[f"{name}.processor"] = m.processor
        return procs

    def set_attn_processor(self, processor):
        """dict keyed like ``attn_processors`` or a single processor (models/unet_2d_condition.py:620-648)."""
        attns = {f"{n}.processor": m for n, m in self.named_modules() if isinstance(m, Attention)}
        if isinstance(processor, dict):
            if len(processor) != len(attns):
                raise ValueError(f"A dict of processors was passed, but the number of processors {len(processor)} does not match "
                                 f"the number of attention layers: {len(attns)}. Please make sure to pass {len(attns)} processor classes.")
            for k, m in attns.items():
                m.set_processor(processor[k])
        else:
            for m in attns.values():
                m.set_processor(processor)

    def set_default_attn_processor(self):
        self.set_attn_processor(AttnProcessor())

    def register_conditioning(self, encoder_hidden_states):
        """Opt-in cache for EAGER loops (``for t in timesteps: unet(x, t, enc)`` — the reference's loop shape, also the forward
        of ``latent_backward_guidance``): project the step-invariant text / image K and V^T of ``encoder_hidden_states`` for
        every IP cross-attention layer once, into buffers owned by that tensor OBJECT; calls that pass the same tensor reuse
        them (and re-project if its ``_version`` or the weights changed).  Without this an unregistered tensor is projected
        on every call (safe default: identity by address would alias a fresh ``torch.cat`` at a recycled address).
        ``DenoiseEngine.set_conditioning`` does the same for its static buffer.  Returns the tensor (in ``unet.dtype``,
        contiguous: pass THIS object to the calls)."""
        enc = encoder_hidden_states
        if enc.dtype != self.dtype or not enc.is_contiguous():
            enc = enc.to(self.dtype).contiguous()
        for m in self.modules():
            if isinstance(m, Attention) and isinstance(m.processor, IPAttnProcessor):
                m.processor.register_static(m, enc)
        return enc

    # ---- forward --------------------------------------------------------------------------------------
    def _timestep_dev(self, timestep, B):
        """-> (device fp32 tensor, stride) for the embedding kernel; Python numbers are cached per value."""
        dev = self.device
        if isinstance(timestep, DeviceSchedule):
            return timestep.table, 0
        if torch.is_tensor(timestep):
            t = timestep
            if t.device != dev or t.dtype != torch.float32:
                t = t.to(device=dev, dtype=torch.float32)
            t = t.reshape(-1)
            return t, (1 if t.numel() == B and B > 1 else 0)
        key = float(timestep)
        if key not in self._t_cache:
            self._t_cache[key] = torch.tensor([key], dtype=torch.float32, device=dev)
        return self._t_cache[key], 0

    def _tproj_weight(self):
        ws = [r.time_emb_proj.weight for r in self._resnets]
        bs = [r.time_emb_proj.bias for r in self._resnets]
        return self._p.get("tproj", ws + bs, lambda: (torch.cat([w.detach() for w in ws], 0).contiguous(),
                                                       torch.cat([b.detach() for b in bs], 0).contiguous()))

    def time_embed(self, timestep, B, added_cond_kwargs=None):
        cfg = self.config
        dt = self.dtype
        t, stride = self._timestep_dev(timestep, B)
        index = timestep.index if isinstance(timestep, DeviceSchedule) else None
        t_emb = ops.timestep_embedding(t, B, cfg.block_out_channels[0], cfg.flip_sin_to_cos, cfg.freq_shift, dt, t_stride=stride,
                                       index=index)
        emb = self.time_embedding.run(t_emb)
        if cfg.addition_embed_type == "text_time":
            if added_cond_kwargs is None or "text_embeds" not in added_cond_kwargs or "time_ids" not in added_cond_kwargs:
                raise ValueError("addition_embed_type 'text_time' requires `text_embeds` and `time_ids` in `added_cond_kwargs`")
            text_embeds = added_cond_kwargs["text_embeds"].to(dt)
            time_ids = added_cond_kwargs["time_ids"].to(device=self.device, dtype=torch.float32).contiguous()
            n_ids = time_ids.shape[1]
            d_add = cfg.addition_time_embed_dim
            add_in = torch.empty((B, text_embeds.shape[1] + n_ids * d_add), dtype=dt, device=self.device)
            add_in[:, :text_embeds.shape[1]].copy_(text_embeds)          # memory placement only
            te = ops.timestep_embedding(time_ids.reshape(-1), B * n_ids, d_add, True, 0.0, dt, t_stride=1)
            add_in[:, text_embeds.shape[1]:].copy_(te.reshape(B, n_ids * d_add))
            aug = ops.linear(add_in, self.add_embedding.linear_1.weight, self.add_embedding.linear_1.bias, act=ops.ACT_SILU)
            emb = ops.linear(aug, self.add_embedding.linear_2.weight, self.add_embedding.linear_2.bias, res=emb)
        w, b = self._tproj_weight()
        tproj = ops.linear(ops.act(emb, ops.ACT_SILU), w, b)            # [B, sum(cout)]: all ResBlock time projections
        return emb, tproj

    def _residual(self, act: _Act, extra):
        """ControlNet / adapter residual added to a token-major activation: an NCHW tensor (diffusers module) or a
        token-major ``_Act`` handed over by ``theatergen_amd.controlnet.ControlNetModel(token_major=True)``."""
        if isinstance(extra, _Act):
            if (extra.b, extra.h, extra.w, extra.c) != (act.b, act.h, act.w, act.c):
                raise ValueError("ControlNet residual geometry mismatch")
            return _Act(ops.add(act.t, extra.t), act.b, act.h, act.w, act.c)
        e = extra.to(act.t.dtype).contiguous()
        e_tok = ops.transpose(e, act.b, act.c, act.hw).reshape(act.b * act.hw, act.c)
        return _Act(ops.add(act.t, e_tok), act.b, act.h, act.w, act.c)

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, timestep_cond=None, attention_mask=None,
                cross_attention_kwargs=None, added_cond_kwargs=None, down_block_additional_residuals=None,
                mid_block_additional_residual=None, encoder_attention_mask=None, return_dict=True, out_dtype=None, time_proj=None):
        """``time_proj`` (optional, [B, sum(cout)] storage dtype): the fused ResBlock time projections of THIS timestep, precomputed by the caller with
        ``time_embed`` (``DenoiseEngine`` keeps one row per step of its schedule: the embedding depends on the timestep only) — the five launches of
        ``time_embed`` are skipped."""
        cfg = self.config
        if attention_mask is not None or encoder_attention_mask is not None or class_labels is not None or timestep_cond is not None:
            raise NotImplementedError("attention masks / class labels / timestep_cond are not on the TheaterGen hot path")
        if not sample.is_cuda:
            raise RuntimeError("theatergen_amd UNet runs on the GPU only (no CPU fallback)")
        dt = self.dtype
        B, _, H, W = sample.shape
        enc = encoder_hidden_states
        if enc.dtype != dt:
            enc = enc.to(dt)
        enc = enc.contiguous()
        ca_kwargs = {} if cross_attention_kwargs is None else cross_attention_kwargs
        track_keys = cross_attention_kwargs is not None

        if time_proj is not None:
            if tuple(time_proj.shape) != (B, self._tproj_width) or time_proj.dtype != dt or time_proj.stride(1) != 1:
                raise ValueError(f"time_proj must be a [{B}, {self._tproj_width}] {dt} tensor of time_embed(...)[1]")
            tproj = time_proj
        else:
            _, tproj = self.time_embed(timestep, B, added_cond_kwargs)

        w_in = self._p.get("conv_in", [self.conv_in.weight], lambda: pack_conv3x3(self.conv_in.weight.detach()))
        x = _Act(ops.conv_in(sample.contiguous(), w_in, self.conv_in.bias, cfg.block_out_channels[0], dt), B, H, W,
                 cfg.block_out_channels[0])

        res = [x]
        for i, blk in enumerate(self.down_blocks):
            for j, resnet in enumerate(blk.resnets):
                x = resnet.run(x, None, tproj)
                if blk.has_cross_attention:
                    if track_keys:
                        ca_kwargs["attn_key"] = ["down", i, j]
                    x = blk.attentions[j].run(x, enc, ca_kwargs)
                res.append(x)
            if blk.downsamplers is not None:
                x = blk.downsamplers[0].run(x)
                res.append(x)

        is_controlnet = mid_block_additional_residual is not None and down_block_additional_residuals is not None
        is_adapter = mid_block_additional_residual is None and down_block_additional_residuals is not None
        if is_adapter:
            raise NotImplementedError("T2I-Adapter residuals are not on the TheaterGen hot path (SD-1.5 flow uses ControlNet)")
        if is_controlnet:                                            # models/unet_2d_condition.py:938-946
            if len(down_block_additional_residuals) != len(res):
                raise ValueError("down_block_additional_residuals length mismatch")
            res = [self._residual(r, e) for r, e in zip(res, down_block_additional_residuals)]

        x = self.mid_block.resnets[0].run(x, None, tproj)
        if track_keys:
            ca_kwargs["attn_key"] = ["mid", 0, 0]
        x = self.mid_block.attentions[0].run(x, enc, ca_kwargs)
        x = self.mid_block.resnets[1].run(x, None, tproj)
        if mid_block_additional_residual is not None:
            x = self._residual(x, mid_block_additional_residual)

        for i, blk in enumerate(self.up_blocks):
            for j, resnet in enumerate(blk.resnets):
                skip = res.pop()
                x = resnet.run(x, skip, tproj)
                if blk.has_cross_attention:
                    if track_keys:
                        ca_kwargs["attn_key"] = ["up", i, j]
                    x = blk.attentions[j].run(x, enc, ca_kwargs)
            if blk.upsamplers is not None:
                x = blk.upsamplers[0].run(x)

        w_out = self._p.get("conv_out", [self.conv_out.weight], lambda: pack_conv3x3(self.conv_out.weight.detach()))
        odt = out_dtype if out_dtype is not None else dt
        if _CONV_OUT_GN and ops.conv_out_takes_gn(x.t.shape[-1], x.h, x.w, cfg.out_channels):
            # conv_norm_out + SiLU inside conv_out's window staging: statistics launch only, the normalised tensor never exists
            coef = ops.groupnorm_coef(x.t, x.b, x.hw, cfg.norm_num_groups, cfg.norm_eps, self.conv_norm_out.weight, self.conv_norm_out.bias)
            out = ops.conv_out(x.t, w_out, self.conv_out.bias, B, x.h, x.w, cfg.out_channels, odt, coef=coef, silu=True)
        else:
            y = ops.groupnorm(x.t, x.b, x.hw, cfg.norm_num_groups, cfg.norm_eps, self.conv_norm_out.weight, self.conv_norm_out.bias,
                              silu=True)
            out = ops.conv_out(y, w_out, self.conv_out.bias, B, x.h, x.w, cfg.out_channels, odt)
        if not return_dict:
            return (out,)
        return UNet2DConditionOutput(sample=out)

    __call__ = forward  # nn.Module hooks are not needed; keeps launch overhead down

    @classmethod
    def from_state_dict(cls, config, state_dict, device="cuda", dtype=torch.bfloat16, ip_adapter=True, num_tokens=4, ip_scale=1.0):
        """Build, optionally install IP-Adapter processors (so ``...attn2.processor.to_k_ip.weight`` keys load), load, move."""
        return _load_on_meta(lambda: _with_ip(cls(config), ip_adapter, num_tokens, ip_scale), state_dict, device, dtype)


def _with_ip(m, ip_adapter, num_tokens, ip_scale):
    if ip_adapter:
        install_ip_processors(m, num_tokens=num_tokens, scale=ip_scale)
    return m


def _load_on_meta(build, state_dict, device, dtype):
    """Construct the module tree on the META device (no 860 M-parameter default init on the host: 27 s for the SD-1.5 UNet), take the state dict's tensors as the
    parameters (``assign=True``), move / convert.  Parameters never alias the caller's state dict."""
    with torch.device("meta"):
        m = build()
    missing, unexpected = m.load_state_dict(state_dict, strict=False, assign=True)
    if unexpected or missing:
        raise RuntimeError(f"state dict mismatch: missing {missing[:5]}... unexpected {unexpected[:5]}...")
    left = [n for n, t in list(m.named_parameters()) + list(m.named_buffers()) if t.is_meta]
    if left:
        raise RuntimeError(f"state dict does not cover {left[:5]}... (constructed on the meta device)")
    m = m.to(device=device, dtype=dtype)
    src = {t.data_ptr() for t in state_dict.values() if torch.is_tensor(t)}
    with torch.no_grad():
        for t in list(m.parameters()) + list(m.buffers()):
            if t.data_ptr() in src:
                t.data = t.data.clone()
    return m


def install_ip_processors(unet, num_tokens=4, scale=1.0):
    """The name-driven processor table of ``IPAdapter.set_ip_adapter`` (reference ip_adapter/ip_adapter.py:95-119)."""
    cfg = unet.config
    procs = {}
    for name in unet.attn_processors.keys():
        cross_attention_dim = None if name.endswith("attn1.processor") else cfg.cross_attention_dim
        if name.startswith("mid_block"):
            hidden_size = cfg.block_out_channels[-1]
        elif name.startswith("up_blocks"):
            hidden_size = list(reversed(cfg.block_out_channels))[int(name[len("up_blocks.")])]
        elif name.startswith("down_blocks"):
            hidden_size = cfg.block_out_channels[int(name[len("down_blocks.")])]
        if cross_attention_dim is None:
            procs[name] = AttnProcessor()
        else:
            procs[name] = IPAttnProcessor(hidden_size=hidden_size, cross_attention_dim=cross_attention_dim, scale=scale,
                                          num_tokens=num_tokens).to(unet.device, dtype=unet.dtype)
    unet.set_attn_processor(procs)
    return procs

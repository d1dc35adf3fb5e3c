"""d loss / d latents through the UNet for ``latent_backward_guidance`` (reference ``models/pipelines.py:62-128``:
``grad_cond = torch.autograd.grad(loss, [latents])[0]`` where ``loss = compute_ca_lossv3(saved cross-attention maps)``).

The HIP path has no autograd, so the reverse pass is explicit.  Weights are frozen: only input gradients exist, and every
contraction of the backward pass is one of the FORWARD MFMA kernels on a transposed (linear) or tap-flipped and channel-
swapped (conv3x3) copy of the weight; what remains — GroupNorm(+SiLU) / LayerNorm / GEGLU / softmax Jacobians, the 2 x 2 sum
behind the nearest upsample — are the kernels of ``csrc/tg_backward.hip``.  Attention is differentiated the way the reference
computes it (materialised probabilities per (batch, head): ``tg_attn_probs`` -> dP = dO V^T -> ``tg_softmax_bwd_rows`` ->
dQ / dK / dV GEMMs): the path runs on ONE image a few times per guided step, not in the 50-step loop.

Mechanics: the forward pass of the blocks up to the last saved attention key runs on the same kernels as inference and records
a tape of (output, inputs, input-gradient closure); the closures capture the few tensors a Jacobian needs (GroupNorm /
LayerNorm inputs, GEGLU pre-activations, the attention block's input).  The loss gradient enters at the saved cross-attention
maps (``compute_ca_lossv3(..., return_grads=True)`` = analytic d loss / d A) as the ``extra`` term of the softmax backward.
PyTorch moves data only (concat, head split / merge, zero-stuffing, transposes): no arithmetic.
"""
import contextlib
import math
import contextvars
import os

import torch

from . import ops
from .attention_processor import AttnProcessor, IPAttnProcessor, tensor_version
from .unet import BasicTransformerBlock, DeviceSchedule, _Act  # noqa: F401
from .weights_pack import pack_conv1x1, pack_conv3x3


FLASH_BWD = os.environ.get("TG_FLASH_BWD", "1") != "0"     # dev A/B knob: 0 = the materialised per-(item, head) reverse pass for every layer


class _StopForward(Exception):
    pass


def _r8(n):
    return (n + 7) // 8 * 8


class Tape:
    def __init__(self):
        self.nodes = []

    def add(self, out, inputs, fn):
        self.nodes.append((out, inputs, fn))
        return out

    def backward(self, target):
        """reverse sweep; returns the gradient accumulated for ``target`` (a forward input tensor)"""
        grads = {}
        for out, inputs, fn in reversed(self.nodes):
            g = grads.pop(id(out), None)
            res = fn(g)
            if res is None:
                continue
            for t, gt in zip(inputs, res):
                if gt is None or t is None:
                    continue
                k = id(t)
                grads[k] = gt if k not in grads else ops.add(grads[k], gt)
        return grads.get(id(target))


# ---- weight views of the backward contractions (cached next to the forward packs) --------------------------------------
def _lin_t(holder, name, w2d):
    """forward out = x W^T with W [N, K]  ->  dx = dy W = tg_gemm(dy, W' = W^T as [K, N])"""
    return holder._p.get("bwd_" + name, [w2d], lambda: w2d.detach().t().contiguous())


def _conv_t(holder, name, w4d):
    """forward conv3x3(x, W [cout, cin, 3, 3], pad 1)  ->  dx = conv3x3(dy, W'[cin, cout, 2 - ky, 2 - kx])"""
    return holder._p.get("bwd_" + name, [w4d], lambda: pack_conv3x3(w4d.detach().permute(1, 0, 2, 3).flip(2, 3).contiguous()))


def _dgrad_lin(g, wt):
    M = g.shape[0]
    return ops.gemm(g, wt, M, wt.shape[0], wt.shape[1])


# ---- attention backward ---------------------------------------------------------------------------------------------------
_SIDE = {}


# per-context side-stream count for ``GraphedInputGrad`` (it used to mutate the process-global os.environ around the capture: not
# thread-safe, and it leaked into eager ``loss_and_grad`` calls of other threads); 0 = fall back to TG_BWD_STREAMS
_STREAMS_OVERRIDE = contextvars.ContextVar("tg_bwd_streams", default=0)


def _side_streams(dev):
    """Side streams of the per-head loops (TG_BWD_STREAMS; default 1 = everything on the current stream: eager launches are bound by the
    host's Python and stream switches add to it — 44.9 -> 50.9 ms per iteration at 768^2 with 4; ``GraphedInputGrad`` captures with 8:
    35.7 -> 31.1 ms per replay)."""
    n = _STREAMS_OVERRIDE.get() or int(os.environ.get("TG_BWD_STREAMS", "1"))
    if n <= 1:
        return []
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), n)
    if key not in _SIDE:
        _SIDE[key] = [torch.cuda.Stream(device=dev) for _ in range(n)]
    return _SIDE[key]


@contextlib.contextmanager
def _on_stream(stream, slot):
    with torch.cuda.stream(stream), ops.workspace_slot(slot):
        yield


def _heads(t2d, B, n, heads, d):
    return t2d.reshape(B, n, heads, d).permute(0, 2, 1, 3).contiguous()          # [B, heads, n, d]


def _pad_rows(t, rows):
    """[.., n, d] -> [.., rows, d] zero-padded (GEMM contraction lengths must be multiples of 8)"""
    if t.shape[-2] == rows:
        return t
    out = torch.zeros((*t.shape[:-2], rows, t.shape[-1]), dtype=t.dtype, device=t.device)
    out[..., :t.shape[-2], :] = t
    return out


def attention_input_grad(attn, proc, h2d, B, N, enc, dout, extra):
    """Gradient of ``to_out(attention(to_q(h), K, V))`` wrt ``h`` [B*N, C].  Self-attention: K, V from ``h`` too; cross-attention:
    K, V (and the IP-Adapter image segment) are constants.  ``dout`` [B*N, C] or None; ``extra`` fp32 [B, heads, N, L_text] or None
    = d loss / d (text attention probabilities) from the guidance loss."""
    from .attention_processor import attn_dims
    inner, heads, d = attn_dims(attn)
    dt, dev = h2d.dtype, h2d.device
    if dout is None and extra is None:
        return None
    # the recompute below is the bias-free, un-normalised, un-masked attention of the SD / SDXL UNets; the forward processors also honour q / k / v
    # biases, attn.group_norm and norm_cross (round 4) — a layer that uses them must not get a silently different gradient
    if any(getattr(l, "bias", None) is not None for l in (attn.to_q, attn.to_k, attn.to_v)) or getattr(attn, "group_norm", None) is not None \
            or getattr(attn, "norm_cross", None) or getattr(attn, "spatial_norm", None) is not None:
        raise NotImplementedError("theatergen_amd.backward: the attention reverse pass covers bias-free projections without group_norm / norm_cross "
                                  "(every SD-1.5 / SD-2.1 / SDXL UNet attention); this layer has one of them")
    if not hasattr(attn, "_p"):
        from .unet import _Packed
        attn._p = _Packed()
    if dout is not None:
        do = _dgrad_lin(dout, _lin_t(attn, "o", attn.to_out[0].weight))
    else:
        do = torch.zeros((B * N, inner), dtype=dt, device=dev)
    q = ops.linear(h2d, attn.to_q.weight)
    self_attn = enc is None
    segs = []                                                        # (k2d, v2d, L, weight, extra)
    if self_attn:
        segs.append((ops.linear(h2d, attn.to_k.weight), ops.linear(h2d, attn.to_v.weight), N, 1.0, None))
    else:
        Ltot, ctx = enc.shape[1], enc.shape[2]
        if isinstance(proc, IPAttnProcessor):
            T = proc.num_tokens
            L = Ltot - T
            et = enc[:, :L].contiguous().reshape(B * L, ctx)
            ei = enc[:, L:].contiguous().reshape(B * T, ctx)
            segs.append((ops.linear(et, attn.to_k.weight), ops.linear(et, attn.to_v.weight), L, 1.0, extra))
            segs.append((ops.linear(ei, proc.to_k_ip.weight), ops.linear(ei, proc.to_v_ip.weight), T, float(proc.scale), None))
        else:
            e2 = enc.contiguous().reshape(B * Ltot, ctx)
            segs.append((ops.linear(e2, attn.to_k.weight), ops.linear(e2, attn.to_v.weight), Ltot, 1.0, extra))
    if self_attn and FLASH_BWD and ops.attention_bwd_supported(d, N):
        # recompute-based reverse pass (round 5, tg_attention_bwd): no N x N matrix, three launches for all (item, head) pairs
        k2d, v2d = segs[0][0], segs[0][1]
        dq, dk, dv = ops.attention_bwd(q, k2d, v2d, do.contiguous(), B, N, heads, d, attn.scale)
        dx = _dgrad_lin(dq, _lin_t(attn, "q", attn.to_q.weight))
        wk, wv = _lin_t(attn, "k", attn.to_k.weight), _lin_t(attn, "v", attn.to_v.weight)
        dx = ops.gemm(dk, wk, B * N, wk.shape[0], wk.shape[1], res=dx)
        dx = ops.gemm(dv, wv, B * N, wv.shape[0], wv.shape[1], res=dx)
        return dx
    if not self_attn and FLASH_BWD and d % 8 == 0 and d <= 64:
        # cross-attention: constant K / V, one softmax per segment (text, IP-Adapter image keys): statistics + dQ launches per segment (tg_attention_bwd_cross)
        dq = None
        doc = do.contiguous()
        for k2d, v2d, L, wgt, ex in segs:
            if wgt == 0.0:
                continue
            part = ops.attention_bwd_cross(q, doc, k2d, v2d, B, N, L, heads, d, attn.scale, attn.scale * wgt,
                                           extra=ex.contiguous() if ex is not None else None)
            dq = part if dq is None else dq + part
        if dq is None:
            dq = torch.zeros_like(q)
        return _dgrad_lin(dq, _lin_t(attn, "q", attn.to_q.weight))
    qh, doh = _heads(q, B, N, heads, d), _heads(do, B, N, heads, d)
    dqh = torch.zeros((B, heads, N, d), dtype=dt, device=dev)
    Np = _r8(N)
    dkh = dvh = None
    if self_attn:
        dkh = torch.empty((B, heads, N, d), dtype=dt, device=dev)
        dvh = torch.empty((B, heads, N, d), dtype=dt, device=dev)
        qT = _pad_rows(qh, Np).transpose(2, 3).contiguous()           # [B, heads, d, Np]
        doT = _pad_rows(doh, Np).transpose(2, 3).contiguous()
    for k2d, v2d, L, wgt, ex in segs:
        if wgt == 0.0:
            continue
        Lp = _r8(L)
        # probabilities: up to 256 keys (every cross-attention, small self-attention maps) straight from tg_attn_probs in fp32;
        # longer self-attention rows as scores GEMM -> tg_softmax_rows per (batch, head), in the storage dtype
        small = L <= 256
        if not small and L % 8 != 0:
            raise RuntimeError("attention backward: more than 256 keys needs a key count that is a multiple of 8")
        probs = ops.attn_probs(q, inner, N * inner, k2d, inner, L * inner, B, 0, heads, d, N, L, attn.scale) if small else None
        kh, vh = _pad_rows(_heads(k2d, B, L, heads, d), Lp), _pad_rows(_heads(v2d, B, L, heads, d), Lp)   # [B, heads, Lp, d]
        kT = kh.transpose(2, 3).contiguous()                          # [B, heads, d, Lp]
        # the per-(item, head) chains below are independent of one another (they write disjoint slices of dqh / dkh / dvh): round-robin
        # over a few side streams (forked from / joined to the current stream, which is what a hipGraph capture needs too; each with its
        # own scratch slot) so that the small GEMMs / row kernels of several heads overlap on the GPU
        side = _side_streams(dev)
        cur = torch.cuda.current_stream(dev)
        for st_ in side:
            st_.wait_stream(cur)
        for b in range(B):
            for hh in range(heads):
                lane_ = (b * heads + hh) % len(side) if side else -1
                with (_on_stream(side[lane_], 8 + lane_) if side else contextlib.nullcontext()):
                    if small:
                        P = probs[b, hh]                                                              # fp32 [N, L]
                    else:
                        P = ops.gemm(qh[b, hh], kh[b, hh], N, L, d)                                   # Q K^T        [N, L]
                        ops.softmax_rows(P, scale=attn.scale, out=P)
                    dP = ops.gemm(doh[b, hh], vh[b, hh], N, Lp, d)                                    # dO V^T       [N, Lp]
                    res = ops.softmax_bwd_rows(P, dP, L, attn.scale * wgt, Lp, extra=ex[b, hh] if ex is not None else None,
                                               want_probs=self_attn)
                    dS, Pst = res if self_attn else (res, None)
                    ops.gemm(dS, kT[b, hh], N, d, Lp, res=dqh[b, hh], out=dqh[b, hh])                 # dQ += dS K   [N, d]
                    if self_attn:
                        if Np == N and Lp == L and dS.is_contiguous() and Pst.is_contiguous():
                            dST = ops.transpose(dS, 1, N, L)              # tiled HIP transpose (tg_transpose): [N, L] -> [L, N]
                            PT = ops.transpose(Pst, 1, N, L)
                        else:                                             # row counts that need zero padding to a multiple of 8
                            dST = torch.zeros((L, Np), dtype=dt, device=dev)
                            dST[:, :N] = dS[:, :L].t()
                            PT = torch.zeros((L, Np), dtype=dt, device=dev)
                            PT[:, :N] = Pst[:, :L].t()
                        ops.gemm(dST, qT[b, hh], L, d, Np, out=dkh[b, hh])                            # dK = dS^T Q  [L, d]
                        ops.gemm(PT, doT[b, hh], L, d, Np, out=dvh[b, hh])                            # dV = P^T dO  [L, d]
        for st_ in side:
            cur.wait_stream(st_)

    def merge(th):
        return th.permute(0, 2, 1, 3).reshape(B * N, inner).contiguous()

    dx = _dgrad_lin(merge(dqh), _lin_t(attn, "q", attn.to_q.weight))
    if self_attn:
        wk, wv = _lin_t(attn, "k", attn.to_k.weight), _lin_t(attn, "v", attn.to_v.weight)
        dx = ops.gemm(merge(dkh), wk, B * N, wk.shape[0], wk.shape[1], res=dx)
        dx = ops.gemm(merge(dvh), wv, B * N, wv.shape[0], wv.shape[1], res=dx)
    return dx


# ---- forward with a tape ------------------------------------------------------------------------------------------------------
class UNetInputGrad:
    """``grad(latents)`` of a scalar function of the saved cross-attention maps of ``unet``."""

    def __init__(self, unet):
        self.unet = unet

    # -- blocks
    def _resnet(self, tape, r, x, skip, tproj):
        b, h, w = x.b, x.h, x.w
        if skip is not None:
            c0 = x.c
            xin = tape.add(torch.cat([x.t, skip.t], dim=1), [x.t, skip.t],
                           lambda g, c0=c0: None if g is None else [g[:, :c0].contiguous(), g[:, c0:].contiguous()])
        else:
            xin = x.t
        cin, cout = r.in_channels, r.out_channels
        if r.output_scale_factor != 1.0:
            raise NotImplementedError("output_scale_factor != 1 is not used by the SD UNets")
        w1 = r._p.get("c1", [r.conv1.weight], lambda: pack_conv3x3(r.conv1.weight.detach()))
        w2 = r._p.get("c2", [r.conv2.weight], lambda: pack_conv3x3(r.conv2.weight.detach()))
        n1, n2 = r.norm1, r.norm2
        h1 = tape.add(ops.groupnorm(xin, b, h * w, r.groups, r.eps, n1.weight, n1.bias, silu=True), [xin],
                      lambda g, xin=xin: None if g is None else [ops.groupnorm_bwd(xin, g, b, h * w, r.groups, r.eps, n1.weight, n1.bias, silu=True)])
        off, width = r.temb_slot
        w1t = _conv_t(r, "c1", r.conv1.weight)
        c1 = tape.add(ops.conv3x3(h1, w1, b, h, w, cin, bias=r.conv1.bias, bvec=tproj[:, off:off + width], rows_per_batch=h * w), [h1],
                      lambda g: None if g is None else [ops.conv3x3(g, w1t, b, h, w, cout)])
        h2 = tape.add(ops.groupnorm(c1, b, h * w, r.groups, r.eps, n2.weight, n2.bias, silu=True), [c1],
                      lambda g, c1=c1: None if g is None else [ops.groupnorm_bwd(c1, g, b, h * w, r.groups, r.eps, n2.weight, n2.bias, silu=True)])
        if r.conv_shortcut is not None:
            ws = r._p.get("sc", [r.conv_shortcut.weight], lambda: pack_conv1x1(r.conv_shortcut.weight.detach()))
            wst = _lin_t(r, "sc", ws)
            res = tape.add(ops.gemm(xin, ws, b * h * w, cout, cin, bias=r.conv_shortcut.bias), [xin],
                           lambda g: None if g is None else [_dgrad_lin(g, wst)])
        else:
            res = xin
        w2t = _conv_t(r, "c2", r.conv2.weight)
        out = tape.add(ops.conv3x3(h2, w2, b, h, w, cout, bias=r.conv2.bias, res=res), [h2, res],
                       lambda g: None if g is None else [ops.conv3x3(g, w2t, b, h, w, cout), g])
        return _Act(out, b, h, w, cout)

    def _transformer(self, tape, tf, x, enc, ca_kwargs, base_key, want_keys, state):
        b, n = x.b, x.hw
        nm = tf.norm
        y = tape.add(ops.groupnorm(x.t, b, n, tf.groups, 1e-6, nm.weight, nm.bias, silu=False), [x.t],
                     lambda g: None if g is None else [ops.groupnorm_bwd(x.t, g, b, n, tf.groups, 1e-6, nm.weight, nm.bias, silu=False)])
        win = tf._w("in", tf.proj_in)
        wint = _lin_t(tf, "in", win)
        y = tape.add(ops.linear(y, win, tf.proj_in.bias), [y], lambda g: None if g is None else [_dgrad_lin(g, wint)])
        for li, blk in enumerate(tf.transformer_blocks):
            y = self._block(tape, blk, y, b, n, enc, ca_kwargs, tuple(base_key) + (li,), want_keys, state)
        wout = tf._w("out", tf.proj_out)
        woutt = _lin_t(tf, "out", wout)
        out = tape.add(ops.linear(y, wout, tf.proj_out.bias, res=x.t), [y, x.t],
                       lambda g: None if g is None else [_dgrad_lin(g, woutt), g])
        return _Act(out, x.b, x.h, x.w, x.c)

    def _block(self, tape, blk, x2d, b, n, enc, ca_kwargs, key, want_keys, state):
        def ln(norm, t):
            return tape.add(ops.layernorm(t, norm.weight, norm.bias, norm.eps), [t],
                            lambda g: None if g is None else [ops.layernorm_bwd(t, g, norm.weight, norm.eps)])

        def attn_node(attn, h, residual, enc_, key_):
            kw = dict(ca_kwargs)
            kw["attn_key"] = list(key_)
            out = BasicTransformerBlock._call(attn, h, b, n, enc_, residual, kw)
            is_key = enc_ is not None and key_ in want_keys

            def bwd(g):
                extra = state["loss_grads"].get(key_) if is_key else None
                dh = attention_input_grad(attn, attn.processor, h, b, n, enc_, g, extra)
                if dh is None:
                    return None
                return [dh, g]
            tape.add(out, [h, residual], bwd)
            if is_key:
                state["seen"].add(key_)
            return out

        h = ln(blk.norm1, x2d)
        x1 = attn_node(blk.attn1, h, x2d, None, key)
        h = ln(blk.norm2, x1)
        x2 = attn_node(blk.attn2, h, x1, enc, key)
        if state["seen"] >= set(want_keys):
            raise _StopForward()                                # nothing after the last saved map influences the loss
        h = ln(blk.norm3, x2)
        ff = blk.ff
        proj, lin2 = ff.net[0].proj, ff.net[2]
        w1t, w2t = _lin_t(ff, "w1", proj.weight), _lin_t(ff, "w2", lin2.weight)
        hh = tape.add(ops.linear(h, proj.weight, proj.bias), [h], lambda g: None if g is None else [_dgrad_lin(g, w1t)])
        gg = tape.add(ops.geglu(hh), [hh], lambda g: None if g is None else [ops.geglu_bwd(hh, g)])
        return tape.add(ops.linear(gg, lin2.weight, lin2.bias, res=x2), [gg, x2],
                        lambda g: None if g is None else [_dgrad_lin(g, w2t), g])

    def _down(self, tape, ds, x):
        wt = _conv_t(ds, "w", ds.conv.weight)
        y = ds.run(x)

        def bwd(g):
            if g is None:
                return None
            z = torch.zeros((x.b, x.h, x.w, x.c), dtype=g.dtype, device=g.device)
            z[:, ::2, ::2] = g.reshape(x.b, y.h, y.w, x.c)           # zero-stuffing: data movement
            return [ops.conv3x3(z.reshape(x.b * x.h * x.w, x.c), wt, x.b, x.h, x.w, x.c)]
        tape.add(y.t, [x.t], bwd)
        return y

    def _up(self, tape, us, x):
        wt = _conv_t(us, "w", us.conv.weight)
        y = us.run(x)
        tape.add(y.t, [x.t], lambda g: None if g is None else
                 [ops.sumpool2x2(ops.conv3x3(g, wt, x.b, y.h, y.w, x.c), x.b, x.h, x.w)])
        return y

    # -- the pass
    def loss_and_grad(self, sample, timestep, encoder_hidden_states, loss_fn, save_keys, cross_attention_kwargs=None,
                      added_cond_kwargs=None):
        """``loss_fn(saved_attn) -> (loss fp32 device scalar, {key: d loss / d map})``; returns (loss, d loss / d sample fp32 NCHW)."""
        unet, cfg = self.unet, self.unet.config
        dt = unet.dtype
        tape = Tape()
        B, _, H, W = sample.shape
        enc = encoder_hidden_states.to(dt).contiguous()
        saved = {}
        want_keys = [tuple(k) for k in save_keys]
        ca_kwargs = dict(cross_attention_kwargs or {})
        ca_kwargs.update({"save_attn_to_dict": saved, "save_keys": want_keys})
        state = {"loss_grads": {}, "seen": set()}
        emb, tproj = unet.time_embed(timestep, B, added_cond_kwargs)
        w_in = unet._p.get("conv_in", [unet.conv_in.weight], lambda: pack_conv3x3(unet.conv_in.weight.detach()))
        c0 = cfg.block_out_channels[0]
        sample_c = sample.contiguous()
        x0 = ops.conv_in(sample_c, w_in, unet.conv_in.bias, c0, dt)
        w_in_t = unet._p.get("bwd_conv_in", [unet.conv_in.weight],
                             lambda: pack_conv3x3(unet.conv_in.weight.detach().permute(1, 0, 2, 3).flip(2, 3).contiguous()))
        zero_bias = torch.zeros(cfg.in_channels, dtype=dt, device=sample.device)
        tape.add(x0, [sample_c], lambda g: None if g is None else
                 [ops.conv_out(g, w_in_t, zero_bias, B, H, W, cfg.in_channels, torch.float32)])
        x = _Act(x0, B, H, W, c0)
        try:
            res = [x]
            for i, blk in enumerate(unet.down_blocks):
                for j, resnet in enumerate(blk.resnets):
                    x = self._resnet(tape, resnet, x, None, tproj)
                    if blk.has_cross_attention:
                        x = self._transformer(tape, blk.attentions[j], x, enc, ca_kwargs, ("down", i, j), want_keys, state)
                    res.append(x)
                if blk.downsamplers is not None:
                    x = self._down(tape, blk.downsamplers[0], x)
                    res.append(x)
            x = self._resnet(tape, unet.mid_block.resnets[0], x, None, tproj)
            x = self._transformer(tape, unet.mid_block.attentions[0], x, enc, ca_kwargs, ("mid", 0, 0), want_keys, state)
            x = self._resnet(tape, unet.mid_block.resnets[1], x, None, tproj)
            for i, blk in enumerate(unet.up_blocks):
                for j, resnet in enumerate(blk.resnets):
                    x = self._resnet(tape, resnet, x, res.pop(), tproj)
                    if blk.has_cross_attention:
                        x = self._transformer(tape, blk.attentions[j], x, enc, ca_kwargs, ("up", i, j), want_keys, state)
                if blk.upsamplers is not None:
                    x = self._up(tape, blk.upsamplers[0], x)
        except _StopForward:
            pass
        missing = set(want_keys) - set(saved.keys())
        if missing:
            raise RuntimeError(f"guidance keys {sorted(missing)} do not name cross-attention layers of this UNet")
        # plain ``AttnProcessor`` on attn2 mirrors the reference's offload quirk (maps after the first are ``.cpu()`` tensors,
        # attention_processor.py:386-389); the loss kernels and the softmax backward read DEVICE memory: bring every map back
        self._had_cpu_maps = any(not v.is_cuda for v in saved.values())
        saved = {k: (v if v.is_cuda else v.to(sample.device)) for k, v in saved.items()}
        loss, grads = loss_fn(saved)
        for k, gk in grads.items():
            if not gk.is_cuda:
                raise RuntimeError(f"latent_backward_guidance: d loss / d map for key {k} is not a device tensor")
        state["loss_grads"] = {tuple(k): g for k, g in grads.items()}
        grad = tape.backward(sample_c)
        if grad is None:
            grad = torch.zeros(sample.shape, dtype=torch.float32, device=sample.device)
        return loss, grad


class GraphedInputGrad:
    """``UNetInputGrad.loss_and_grad`` captured ONCE into a hipGraph and replayed: for callers that run many iterations with the same
    shapes, conditioning and loss function (the inner loop of ``latent_backward_guidance``: up to 5 iterations x 10 steps per image).
    An eager iteration is ~3000 launches and bound by the host (43 ms at 768^2); the replay is bound by the GPU (31 ms with the per-head
    chains on 8 forked streams), bit-identical.  ``sample`` and ``timestep`` are static device buffers: ``run(sample, timestep)`` copies
    into them.  The loss function must be capturable (``compute_ca_lossv3(return_grads=True)`` is: device-resident plan)."""

    def __init__(self, unet, sample, timestep, encoder_hidden_states, loss_fn, save_keys, streams=8, **kw):
        self.engine = UNetInputGrad(unet)
        self.sample = sample.detach().clone().contiguous()
        self.timestep = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).to(sample.device).clone()
        self.unet = unet
        self._enc_src, self._enc_ver = encoder_hidden_states, tensor_version(encoder_hidden_states)
        # a PRIVATE static copy of the conditioning (the caller's tensor may be a temporary): what the captured launches' K / V^T caches belong to
        self.enc = unet.register_conditioning(encoder_hidden_states.detach().to(unet.dtype).clone().contiguous())
        args = (self.sample, self.timestep, self.enc, loss_fn, save_keys)
        token = _STREAMS_OVERRIDE.set(int(streams))
        try:
            with torch.no_grad():
                self.engine.loss_and_grad(*args, **kw)            # eager once: allocator, packed weights, scratch, the loss plan
                torch.cuda.synchronize()
                if getattr(self.engine, "_had_cpu_maps", False):
                    # the host -> device copy of such a map inside the capture fails with an opaque capture error
                    raise RuntimeError("GraphedInputGrad: a guidance key's cross-attention layer runs the plain AttnProcessor, whose captured maps "
                                       "live on the CPU (reference quirk, attention_processor.py:386-389); install IPAttnProcessor / CNAttnProcessor "
                                       "on the guidance layers or use graphed=False")
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self.loss, self.grad = self.engine.loss_and_grad(*args, **kw)
        finally:
            _STREAMS_OVERRIDE.reset(token)

    def set_conditioning(self, encoder_hidden_states):
        """New embeddings of the captured shape: copied into the graph's static conditioning buffer, K / V^T re-projected IN PLACE (the buffers the
        captured launches read); a no-op for the tensor the graph already holds."""
        src = encoder_hidden_states
        if src is self.enc:
            return
        # same tensor OBJECT as last time: skipped only when it was not written since (a caller that refills one static buffer with
        # copy_ for the next character hands over the same object with new contents; an inference-mode tensor has no version: always refreshed)
        ver = tensor_version(src)
        if src is self._enc_src and ver is not None and ver == self._enc_ver:
            return
        if tuple(src.shape) != tuple(self.enc.shape):
            raise RuntimeError(f"GraphedInputGrad: conditioning of shape {tuple(src.shape)}, captured with {tuple(self.enc.shape)}")
        self.enc.copy_(src.to(self.enc.dtype))
        self.unet.register_conditioning(self.enc)
        self._enc_src, self._enc_ver = src, ver

    def run(self, sample=None, timestep=None):
        """-> (loss, grad): the graph's static output tensors (overwritten by the next ``run``)."""
        if sample is not None:
            self.sample.copy_(sample)
        if timestep is not None:
            self.timestep.copy_(torch.as_tensor(timestep, dtype=torch.float32).reshape(-1))
        self.graph.replay()
        return self.loss, self.grad


def latent_backward_guidance(adapter, scheduler, unet, cond_embeddings, index, bboxes, object_positions, t, latents, loss,
                             loss_scale=30, loss_threshold=0.2, max_iter=5, max_index_step=10, cross_attention_kwargs=None,
                             ref_ca_saved_attns=None, guidance_attn_keys=None, verbose=False, clear_cache=False, prompt_embeds=None,
                             final=False, graphed=False, **kwargs):
    """Reference ``models/pipelines.py:62-128`` with the same signature: while the (de-scaled) loss is above the threshold, run the
    UNet on ``latents`` with the attention-map side channel, evaluate ``compute_ca_lossv3`` times ``loss_scale``, take
    d loss / d latents and step ``latents -= sqrt(1 - alpha_bar_t) * grad`` (the ``alphas_cumprod`` branch :108-115: DDIM has no
    ``sigmas``).  ``loss`` is a Python float or a tensor (the reference passes a large initial value).
    ``graphed=True`` (not a reference argument): the iteration is captured ONCE per (shapes, conditioning tensor, boxes, positions, keys,
    loss arguments) into a hipGraph (``GraphedInputGrad``, cached on the UNet) and replayed for every iteration of every step — the
    timestep is a device buffer of the graph; same bits as the eager loop.  Not available with ``ref_ca_saved_attns`` (per-step reference
    maps are host-side inputs of the loss)."""
    from . import guidance as G
    from .pipelines import DEFAULT_GUIDANCE_ATTN_KEYS
    keys = [tuple(k) for k in (guidance_attn_keys if guidance_attn_keys is not None else DEFAULT_GUIDANCE_ATTN_KEYS)]
    iteration = 0
    if index >= max_index_step:
        return latents, loss
    if isinstance(max_iter, list):
        max_iter = max_iter[index] if len(max_iter) > index else max_iter[-1]
    engine = UNetInputGrad(unet)
    lat32 = latents.detach().to(torch.float32).contiguous()
    if not graphed:
        cond_embeddings = unet.register_conditioning(cond_embeddings)  # K / V^T of the conditioning projected once, not per iteration

    def loss_fn(saved):
        return G.compute_ca_lossv3(saved_attn=saved, bboxes=bboxes, object_positions=object_positions, guidance_attn_keys=keys,
                                   ref_ca_saved_attns=ref_ca_saved_attns, index=index, verbose=verbose, return_grads=True,
                                   loss_scale=float(loss_scale), **kwargs)

    def val(v):
        return float(v.item()) if torch.is_tensor(v) else float(v)

    gig = None
    if graphed:
        if ref_ca_saved_attns is not None or cross_attention_kwargs:
            raise RuntimeError("latent_backward_guidance(graphed=True): ref_ca_saved_attns / cross_attention_kwargs need the eager loop")
        cache = unet.__dict__.setdefault("_graphed_input_grad", {})
        # the key is (shapes, dtypes, loss arguments, a weight / processor epoch) — NOT the identity of the conditioning tensor: a caller
        # that builds a fresh ``torch.cat`` per step gets its embeddings COPIED into the graph's static buffer (K / V^T re-projected in place);
        # an in-place weight update, a LoRA merge or ``set_attn_processor`` changes the epoch and re-captures (the graph bakes packed weights)
        epoch = (sum(tensor_version(p_) or 0 for p_ in unet.parameters()), tuple(id(v) for v in unet.attn_processors.values()))
        gkey = (tuple(latents.shape), str(unet.dtype), tuple(cond_embeddings.shape), str(cond_embeddings.dtype), repr(bboxes), repr(object_positions),
                repr(keys), float(loss_scale), repr(sorted(kwargs.items())), epoch)
        gig = cache.get(gkey)
        if gig is None:
            if cache:
                print("[theatergen_amd] latent_backward_guidance(graphed=True): re-capturing the iteration (shapes / loss arguments / weights changed)")
            cache.clear()                                              # one captured iteration at a time (its buffers are GB-scale at 768^2)
            gig = cache[gkey] = GraphedInputGrad(unet, scheduler.scale_model_input(lat32, t).to(unet.dtype), t, cond_embeddings, loss_fn, keys)
        else:
            gig.set_conditioning(cond_embeddings)

    while val(loss) / loss_scale > loss_threshold and iteration < max_iter and index < max_index_step:
        model_in = scheduler.scale_model_input(lat32, t).to(unet.dtype)
        if gig is not None:
            loss, grad = gig.run(model_in, float(t))
        else:
            loss, grad = engine.loss_and_grad(model_in, t, cond_embeddings, loss_fn, keys, cross_attention_kwargs=cross_attention_kwargs)
        if not math.isfinite(val(loss)):
            print("**Loss is NaN**")
        if hasattr(scheduler, "sigmas"):
            step = float(scheduler.sigmas[index]) ** 2
        elif hasattr(scheduler, "alphas_cumprod"):
            step = float((1 - scheduler.alphas_cumprod[int(t)]) ** 0.5)
        else:
            step = 1.0
        one = torch.ones(1, dtype=torch.float32, device=lat32.device)
        lat32 = ops.add_noise(lat32.reshape(-1), grad.reshape(-1), one, -step * one)[0].reshape(lat32.shape)       # latents - step * grad
        iteration += 1
    if gig is not None and torch.is_tensor(loss):
        loss = loss.clone()                                            # the graph's static output would change under the caller at the next replay
    return lat32.to(latents.dtype), loss

"""Attention module + attention processors with the reference's diffusers call surface, running on HIP.

Mirrors reference ``ip_adapter/attention_processor.py``:
  ``Attention``        :12-279   (parameter names to_q / to_k / to_v / to_out.0; ``heads``; ``scale``)
  ``AttnProcessor``    :282-393  (self / plain cross attention)            -> installed on every ``attn1``
  ``IPAttnProcessor``  :396-553  (decoupled text + image cross attention)  -> installed on every ``attn2``
  ``CNAttnProcessor``  :861-923  (ControlNet: text tokens only)
Same constructor arguments, same mutable ``.scale`` / ``.num_tokens`` attributes (``IPAdapter.set_scale`` mutates
them by ``isinstance`` check, reference ``ip_adapter/ip_adapter.py:155-158``), same ``__call__`` keyword set,
``to_k_ip`` / ``to_v_ip`` parameter names so ``ModuleList(unet.attn_processors.values()).load_state_dict``
(``ip_adapter.py:139-140``) works unchanged.

What differs: the computation.  ``baddbmm -> softmax -> bmm`` with a materialised [B*h, N, Lk] tensor
(:187-219) becomes ONE fused MFMA kernel (``tg_attention``); the head split / merge permutes (:169-185)
disappear because the kernel reads Q / K / V^T straight from the projection GEMM outputs; the two softmaxes
of the IP path stay independent (segment 1 of the kernel).  ``torch.nn`` modules are used as parameter
containers only; there is no PyTorch compute and no CPU fallback.

DROP-IN CONTRACT (INTEGRATION.md level 2): the processors read from ``attn`` ONLY what the reference
``Attention`` (:12-279) and diffusers 0.21.4's carry — ``heads``, ``scale``, ``to_q`` / ``to_k`` / ``to_v`` / ``to_out``,
``group_norm`` / ``spatial_norm`` / ``norm_cross``, ``residual_connection``, ``rescale_output_factor`` — so they run on a FOREIGN
``Attention`` installed through ``unet.set_attn_processor({...})``.  Head geometry is derived from the weights
(``inner = to_q.weight.shape[0]``, ``d = inner // heads``: ``inner_dim`` is a local of the reference constructor, :52); the packed
``q|k|v`` / ``k|v`` / LayerNorm-folded weights live in a PROCESSOR-SIDE cache keyed by the ``attn`` object (weak reference) and the
parameters' ``_version`` / storage, never on the module.
"""
import os
import weakref

import torch
import torch.nn as nn

from . import ops, rowchain


def _round8(n):
    return (n + 7) // 8 * 8


def tensor_version(t):
    """``t._version`` or None where it is unavailable (inference-mode tensors raise): None never compares equal to a
    recorded version, so such tensors are simply re-projected."""
    try:
        return t._version
    except RuntimeError:
        return None


# ---- streams that REPLAY captured steps reading device scalars written from the host side (the IP scale) ------------------------------------------
# ``DenoiseEngine.run_concurrent`` replays each engine's hipGraph on an engine-owned stream.  A 4-byte fill issued on the CURRENT stream is not
# ordered against kernels already queued on those streams (VERDICT r4 weak item 5 / ADVICE r3): ``IPAttnProcessor.scale = s`` therefore fences the
# fill against every registered replay stream — the current stream first waits for what they have queued (no write under a running reader), they
# then wait for the fill (every later replay reads the new value).  No registered stream (the default single-stream engine): no cost.
_REPLAY_STREAMS = []          # (torch.device, torch.cuda.Stream): STRONG references — the owner unregisters (``unregister_replay_stream``, a
                              # ``weakref.finalize`` of the engine); a weakly referenced Stream object crashed in ``.device`` on the GPU box


def register_replay_stream(stream, device=None):
    dev = torch.device(device) if device is not None else stream.device
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    for d, st in _REPLAY_STREAMS:
        if st is stream:
            return stream
    _REPLAY_STREAMS.append((dev, stream))
    return stream


def unregister_replay_stream(stream):
    _REPLAY_STREAMS[:] = [(d, st) for d, st in _REPLAY_STREAMS if st is not stream]


def _live_replay_streams(device):
    return [st for d, st in _REPLAY_STREAMS if d == device]


def fenced_fill(t, value):
    """``t.fill_(value)`` on the current stream, ordered after everything queued on the registered replay streams and before whatever they run next"""
    dev = t.device
    if dev.type != "cuda" or torch.cuda.is_current_stream_capturing():
        t.fill_(value)
        return
    cur = torch.cuda.current_stream(dev)
    others = [st for st in _live_replay_streams(dev) if st != cur]
    for st in others:
        cur.wait_stream(st)
    t.fill_(value)
    for st in others:
        st.wait_stream(cur)


class StaticSlots:
    """Step-invariant buffers derived from a conditioning tensor (text / image K and V^T, the ControlNet conditioning
    embedding) are cached ONLY for tensors their owner registered explicitly (``DenoiseEngine``'s static buffers).

    Identity is the tensor OBJECT, held by weak reference — never its address: a fresh ``torch.cat`` of the next
    character's embeddings that the caching allocator places at the freed address of the previous one (same shape,
    ``_version`` 0 again: exactly the reference's usage, ``models/pipelines.py:230-233, 860-950``) is a different object
    and is projected anew.  A slot lives as long as its tensor: nothing is evicted while a captured hipGraph of the
    owner may still read the buffers, and the buffers are released when the owner drops the tensor."""

    def __init__(self):
        self._slots = {}

    def get(self, t):
        hit = self._slots.get(id(t))
        return hit[1] if hit is not None and hit[0]() is t else None

    def put(self, t, payload):
        key, slots = id(t), self._slots

        def _drop(ref, key=key, slots=slots):
            cur = slots.get(key)
            if cur is not None and cur[0] is ref:
                del slots[key]

        slots[key] = (weakref.ref(t, _drop), payload)
        return payload

    def __len__(self):
        return len(self._slots)


# ---- processor-side packed-weight cache -----------------------------------------------------------------------------------
_PACKS = weakref.WeakKeyDictionary()          # attn module (any class) -> {name: (key, payload)}


def _cached(attn, name, tensors, build):
    """``build()`` once per (attn object, name) and again whenever one of ``tensors`` was written in place (``_version``),
    re-allocated, cast or moved.  The entry dies with the module."""
    try:
        packs = _PACKS.get(attn)
        if packs is None:
            packs = _PACKS[attn] = {}
    except TypeError:                           # an attn object that cannot be weakly referenced: no caching
        packs = {}
    key = tuple((t.data_ptr(), tensor_version(t), t.dtype, t.device) for t in tensors)
    hit = packs.get(name)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            hit = (key, build())
        packs[name] = hit
    return hit[1]


def attn_dims(attn):
    """(inner, heads, head_dim) of any diffusers-shaped ``Attention``: from the projection weight, not from attributes the reference
    class does not have."""
    inner = attn.to_q.weight.shape[0]
    heads = int(attn.heads)
    if inner % heads or (inner // heads) % 8:
        raise RuntimeError(f"theatergen_amd: to_q has {inner} rows for {heads} heads; head_dim must be a multiple of 8")
    return inner, heads, inner // heads


def _cat_bias(lins, dtype_like):
    """concatenated q|k|v bias (None when no projection has one; zeros for the ones that lack it)"""
    if all(getattr(l, "bias", None) is None for l in lins):
        return None
    return torch.cat([l.bias.detach() if l.bias is not None else torch.zeros(l.weight.shape[0], dtype=dtype_like.dtype, device=dtype_like.device)
                      for l in lins]).to(dtype_like.dtype).contiguous()


def qkv_weight(attn):
    """([3*inner, C] rows = to_q ; to_k ; to_v, bias or None): self-attention in one GEMM."""
    lins = [attn.to_q, attn.to_k, attn.to_v]
    ws = [l.weight for l in lins]
    bs = [l.bias for l in lins if l.bias is not None]
    return _cached(attn, "qkv", ws + bs, lambda: (torch.cat([w.detach() for w in ws], dim=0).contiguous(), _cat_bias(lins, ws[0])))


def kv_weight(attn):
    lins = [attn.to_k, attn.to_v]
    ws = [l.weight for l in lins]
    bs = [l.bias for l in lins if l.bias is not None]
    return _cached(attn, "kv", ws + bs, lambda: (torch.cat([w.detach() for w in ws], dim=0).contiguous(), _cat_bias(lins, ws[0])))


def ln_weight(attn, name, norm):
    """(W', u, v) of ``pack_ln_linear`` for the projection that consumes ``norm`` (a ``nn.LayerNorm``): ``name`` = "qkv" (self-
    attention: to_q ; to_k ; to_v rows) or "q" (cross-attention query)."""
    from .weights_pack import pack_ln_linear
    lins = [attn.to_q, attn.to_k, attn.to_v] if name == "qkv" else [attn.to_q]
    ws = [l.weight for l in lins]
    bs = [l.bias for l in lins if l.bias is not None]
    return _cached(attn, "ln_" + name, ws + bs + [norm.weight, norm.bias],
                   lambda: pack_ln_linear(torch.cat([w.detach() for w in ws], dim=0), _cat_bias(lins, ws[0]), norm.weight, norm.bias))


# ---- round 5: norm2 + to_q + cross-attention as ONE launch on the inner levels (tg_xq_attn: csrc/tg_xattn_epi.h) ------------------------------------------
# TG_XQ=0 switches it off (old three-launch path); TG_XQ_MIN_ROWS: below this many token rows the 128 x 160 tiling leaves most of the chip idle
XQ_ENABLED = os.environ.get("TG_XQ", "1") != "0"
XQ_MIN_ROWS = int(os.environ.get("TG_XQ_MIN_ROWS", "4096"))


def xq_eligible(attn, x2d, B, N, L, T, kwargs):
    """inner-level geometry the fused launch is built for: square to_q with C % 320 == 0, head dim 80 | 160, whole 128-token tiles per batch item,
    <= 96 text and <= 16 image keys; nothing that needs q or the probabilities afterwards (capture, masks)"""
    if not XQ_ENABLED or x2d.dtype not in (torch.bfloat16, torch.float16):
        return False
    inner, heads, d = attn_dims(attn)
    C = x2d.shape[1]
    if inner != C or attn.to_q.weight.shape[1] != C or C % 320 or d not in (80, 160) or N % 128 or B * N < XQ_MIN_ROWS:
        return False
    if L < 1 or L > 96 or T < 0 or T > 16 or x2d.stride(0) != C or x2d.stride(1) != 1:
        return False
    for k in ("save_attn_to_dict", "attention_mask", "attn_process_fn"):
        if kwargs.get(k) is not None:
            return False
    return not kwargs.get("return_attntion_probs")


def xq_weight(attn, norm):
    """(W', u, v): LayerNorm fold of (softmax scale * log2 e) * to_q — the fused launch's exp2 takes the scores as they leave the MFMA"""
    import math
    from .weights_pack import pack_ln_linear
    ts = [attn.to_q.weight, norm.weight, norm.bias] + ([attn.to_q.bias] if attn.to_q.bias is not None else [])
    # the folded scale is part of the key: a caller that changes attn.scale (scale_qk toggles on a foreign Attention) gets a new pack
    return _cached(attn, f"xq_q_{float(attn.scale)!r}", ts, lambda: pack_ln_linear(attn.to_q.weight.detach(), attn.to_q.bias, norm.weight, norm.bias,
                                                            scale=float(attn.scale) * math.log2(math.e)))


class Attention(nn.Module):
    """Parameter container + dispatcher (reference attention_processor.py:12-167).  Carries exactly the reference's attribute set
    (plus ``query_dim`` / ``is_cross`` for this package's UNet); the processors below do not depend on this class."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, cross_attention_norm=None, cross_attention_norm_num_groups=32,
                 added_kv_proj_dim=None, norm_num_groups=None, spatial_norm_dim=None, out_bias=True, scale_qk=True,
                 only_cross_attention=False, eps=1e-5, rescale_output_factor=1.0, residual_connection=False,
                 _from_deprecated_attn_block=False, processor=None):
        super().__init__()
        for name, val in (("added_kv_proj_dim", added_kv_proj_dim), ("spatial_norm_dim", spatial_norm_dim)):
            if val is not None:
                raise ValueError(f"theatergen_amd.Attention: {name} is not on the TheaterGen hot path (unsupported)")
        if only_cross_attention:
            raise ValueError("`only_cross_attention` can only be set to True if `added_kv_proj_dim` is not None.")
        inner_dim = dim_head * heads
        self.query_dim = query_dim
        self.is_cross = cross_attention_dim is not None
        cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.cross_attention_dim = cross_attention_dim
        self.upcast_attention = upcast_attention      # accumulation / softmax are always fp32 in the kernel
        self.upcast_softmax = upcast_softmax
        self.rescale_output_factor = rescale_output_factor
        self.residual_connection = residual_connection
        self._from_deprecated_attn_block = _from_deprecated_attn_block
        self.scale_qk = scale_qk
        self.scale = dim_head ** -0.5 if scale_qk else 1.0
        self.heads = heads
        self.sliceable_head_dim = heads
        self.added_kv_proj_dim = None
        self.only_cross_attention = False
        self.group_norm = nn.GroupNorm(num_channels=query_dim, num_groups=norm_num_groups, eps=eps, affine=True) if norm_num_groups is not None else None
        self.spatial_norm = None
        if cross_attention_norm is None:
            self.norm_cross = None
        elif cross_attention_norm == "layer_norm":
            self.norm_cross = nn.LayerNorm(cross_attention_dim)
        elif cross_attention_norm == "group_norm":
            self.norm_cross = nn.GroupNorm(num_channels=cross_attention_dim, num_groups=cross_attention_norm_num_groups, eps=1e-5, affine=True)
        else:
            raise ValueError(f"unknown cross_attention_norm: {cross_attention_norm}. Should be None, 'layer_norm' or 'group_norm'")
        self.to_q = nn.Linear(query_dim, inner_dim, bias=bias)
        self.to_k = nn.Linear(cross_attention_dim, inner_dim, bias=bias)
        self.to_v = nn.Linear(cross_attention_dim, inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner_dim, query_dim, bias=out_bias), nn.Dropout(dropout)])
        self.set_processor(processor if processor is not None else AttnProcessor())

    def set_processor(self, processor):
        if hasattr(self, "processor") and isinstance(self.processor, nn.Module) and not isinstance(processor, nn.Module):
            self._modules.pop("processor")
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **cross_attention_kwargs)


def _check_common(attn, hidden_states, attn_process_fn):
    if attn_process_fn is not None:
        # reference :350-352 hands the MATERIALISED [B*h, N, L] probabilities to a Python callback; no caller in the reference passes one
        raise NotImplementedError("theatergen_amd: attn_process_fn would need materialised probabilities (unsupported)")
    if getattr(attn, "spatial_norm", None) is not None:
        # reference :316-317 (SpatialNorm of the MoVQ decoder: conditioned on `temb` as a latent image); not built by any SD / SDXL UNet or VAE
        raise NotImplementedError("theatergen_amd: attn.spatial_norm (MoVQ SpatialNorm) is not supported")
    if getattr(attn, "added_kv_proj_dim", None) is not None:
        raise NotImplementedError("theatergen_amd: added_kv_proj_dim belongs to the AttnAddedKV processors (unsupported)")
    _need_gpu(hidden_states)


def _need_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("theatergen_amd: attention runs on the GPU only (no CPU fallback)")


def _group_norm_tokens(gn, x2d, B, N):
    """``attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)`` (reference :330-331) on the token-major [B*N, C] view:
    statistics per (batch item, channel group) over all N tokens."""
    return ops.groupnorm(x2d, B, N, gn.num_groups, gn.eps, gn.weight, gn.bias)


def _norm_encoder(attn, enc):
    """``attn.norm_encoder_hidden_states`` (reference :261-279): LayerNorm over the channel dim, or GroupNorm over (channel group x
    sequence) per batch item.  ``enc`` [B, L, ctx] -> normalised contiguous [B, L, ctx]."""
    nc = attn.norm_cross
    B, L, ctx = enc.shape
    e2 = enc.contiguous().reshape(B * L, ctx)
    if isinstance(nc, nn.LayerNorm):
        return ops.layernorm(e2, nc.weight, nc.bias, nc.eps).reshape(B, L, ctx)
    if isinstance(nc, nn.GroupNorm):
        return ops.groupnorm(e2, B, L, nc.num_groups, nc.eps, nc.weight, nc.bias).reshape(B, L, ctx)
    raise RuntimeError(f"theatergen_amd: unknown attn.norm_cross {type(nc).__name__}")


def _prepare_mask(attention_mask, L, B, heads, N):
    """``attn.prepare_attention_mask`` (reference :221-259, out_dim 3) + the broadcast rules of ``baddbmm(mask, q, k^T)`` (:193-206)
    -> fp32 [Bm, Hm, Qm, L] for ``ops.attention(mask=)``.  Accepted, like the reference: [B, Q, L] (repeated over heads) or
    [B*heads, Q, L], with Q in {1, N}; a mask whose key length is not L is padded by the reference with L more zeros and then fails
    in baddbmm — here it is a RuntimeError up front."""
    m = attention_mask
    if m.ndim != 3:
        raise RuntimeError(f"theatergen_amd: attention_mask must be 3-D [batch(*heads), 1 or queries, keys], got {tuple(m.shape)}")
    if m.shape[-1] != L:
        raise RuntimeError(f"theatergen_amd: attention_mask covers {m.shape[-1]} keys, the key sequence has {L} "
                           "(prepare_attention_mask would pad it to a length baddbmm rejects)")
    if m.shape[1] not in (1, N):
        raise RuntimeError(f"theatergen_amd: attention_mask has {m.shape[1]} query rows for {N} queries")
    m = m.to(torch.float32)
    if m.shape[0] == B * heads and heads > 1:
        m = m.reshape(B, heads, m.shape[1], L)
    elif m.shape[0] == B:
        m = m.reshape(B, 1, m.shape[1], L)
    elif m.shape[0] == 1:
        m = m.reshape(1, 1, m.shape[1], L)
    else:
        raise RuntimeError(f"theatergen_amd: attention_mask batch {m.shape[0]} matches neither batch {B} nor batch * heads {B * heads}")
    return m.contiguous()


def _to_tokens(hidden_states):
    """[B,N,C] (or [B,C,H,W] -> token-major through tg_transpose) -> ([B*N, C] view, B, N, C, shape4)."""
    if hidden_states.ndim == 4:
        b, c, h, w = hidden_states.shape
        x = ops.transpose(hidden_states.contiguous(), b, c, h * w)         # [b, hw, c]
        return x.reshape(b * h * w, c), b, h * w, c, (b, c, h, w)
    b, n, c = hidden_states.shape
    return hidden_states.contiguous().reshape(b * n, c), b, n, c, None


def _finish(attn, o2d, B, N, C, shape4, x_tokens, fused_residual):
    """to_out projection (+bias) with the optional residual / rescale folded into the GEMM epilogue:
    ``(to_out(o) + residual) / rescale_output_factor`` (reference :358-369).  ``x_tokens`` is the token-major
    view of the processor input (= the residual of ``residual_connection``)."""
    res = fused_residual
    if res is None and attn.residual_connection:
        res = x_tokens
    out = None
    if attn.rescale_output_factor == 1.0:
        out = rowchain.linear320(o2d, attn.to_out[0].weight, attn.to_out[0].bias, res, attn, "to_out", _cached)
    if out is None:
        out = ops.linear(o2d, attn.to_out[0].weight, attn.to_out[0].bias, res=res, out_scale=1.0 / attn.rescale_output_factor)
    if shape4 is not None:
        b, c, h, w = shape4
        return ops.transpose(out, b, h * w, c).reshape(b, c, h, w)
    return out.reshape(B, N, C)


def _enc_rows(enc):
    """encoder_hidden_states [B, L, ctx] possibly narrowed along dim 1 (``enc[:, :end]``) -> (tensor, rows/batch, batch pitch)
    for tg_gemm's batched-A addressing; falls back to a contiguous copy for any other layout."""
    B, L, ctx = enc.shape
    if enc.stride(2) == 1 and enc.stride(1) == ctx and enc.stride(0) >= L * ctx and enc.stride(0) % 8 == 0:
        return enc, L, enc.stride(0)
    enc = enc.contiguous()
    return enc, L, L * ctx


def _save_probs(attn, q, q_ld, k, k_ld, B, N, L, save_attn_to_dict, save_keys, attn_key, return_cond_ca_only,
                return_token_ca_only, offload_cross_attn_to_cpu):
    """Attention-map capture side channel (reference :532-551): fp32 [B', heads, N, tokens]."""
    b0 = 0
    if return_cond_ca_only:
        assert B % 2 == 0, f"Samples are not in pairs: {B} samples"
        b0 = B // 2
    tokens = None
    if return_token_ca_only is not None:
        if isinstance(return_token_ca_only, int):
            tokens = torch.tensor([return_token_ca_only], dtype=torch.int32, device=q.device)
        else:
            tokens = torch.as_tensor(return_token_ca_only).to(device=q.device, dtype=torch.int32).reshape(-1)
    _, heads, d = attn_dims(attn)
    probs = ops.attn_probs(q, q_ld, N * q_ld, k, k_ld, L * k_ld, B, b0, heads, d, N, L, attn.scale, tokens)
    if offload_cross_attn_to_cpu:
        probs = probs.cpu()
    if save_attn_to_dict is not None and (save_keys is None or (tuple(attn_key) in save_keys)):
        save_attn_to_dict[tuple(attn_key)] = probs
    return probs


def front_eligible(attn, kwargs):
    """attn1 of a first-level block can take its Q | K | V^T from ``tg_rc_front`` (plain self-attention through our AttnProcessor)"""
    if type(attn.processor) is not AttnProcessor or kwargs.get("attention_mask") is not None or kwargs.get("attn_process_fn") is not None:
        return False
    if getattr(attn, "group_norm", None) is not None or getattr(attn, "spatial_norm", None) is not None:
        return False
    if attn.rescale_output_factor != 1.0 or attn.residual_connection or kwargs.get("return_attntion_probs"):
        return False
    inner, heads, d = attn_dims(attn)
    return inner == 320 and attn.to_q.weight.shape[1] == 320


def self_attention_from_qkv(attn, qk, vt, ldt, B, N, residual):
    """attn1 with its projections already computed (``tg_rc_front``): flash attention over Q | K [B*N, 2 * inner] / V^T + to_out + residual"""
    inner, heads, d = attn_dims(attn)
    o = torch.empty((B * N, inner), dtype=qk.dtype, device=qk.device)
    ops.attention(qk, 2 * inner, N * 2 * inner, qk[:, inner:], 2 * inner, N * 2 * inner, vt, ldt, inner * ldt, N,
                  B, heads, d, N, attn.scale, o, inner, N * inner)
    return _finish(attn, o, B, N, inner, None, None, residual).reshape(B * N, -1)


def fused_cross_block(attn, norm, x2d, B, N, enc, kwargs):
    """norm2 + attn2 + residual of a first-level ``BasicTransformerBlock`` in ONE launch (``tg_rc_xattn``): returns the new stream
    [B*N, 320], or None when the block is not eligible (other geometry, a foreign processor, attention-map capture, masks, ...) and the
    caller runs the processor as usual.  Same arithmetic contract as the three-launch path: LayerNorm statistics in fp32 on the stored
    rows, q rounded to the storage dtype, two independent fp32 softmaxes, O rounded once, fp32 accumulation of to_out + bias + residual."""
    proc = attn.processor
    if type(proc) not in (AttnProcessor, IPAttnProcessor) or enc is None or enc.ndim != 3:
        rowchain.trace("xattn no: processor / enc", type(proc).__name__, None if enc is None else tuple(enc.shape))
        return None
    for k in ("save_attn_to_dict", "attention_mask", "attn_process_fn"):
        if kwargs.get(k) is not None:
            return None
    if kwargs.get("return_attntion_probs") or attn.rescale_output_factor != 1.0 or attn.residual_connection:
        return None
    if getattr(attn, "group_norm", None) is not None or getattr(attn, "norm_cross", None) or getattr(attn, "spatial_norm", None) is not None:
        return None
    is_ip = type(proc) is IPAttnProcessor
    T = int(proc.num_tokens) if is_ip else 0
    L = enc.shape[1] - T
    C = x2d.shape[1]
    if enc.shape[0] != B or not rowchain.xattn_eligible(attn, C, B, N, L, T, x2d.dtype) or x2d.stride(0) != C:
        rowchain.trace("xattn no", tuple(enc.shape), B, N, C, L, T)
        return None
    rowchain.trace("xattn yes", B, N, T)
    inner = 320
    if is_ip:
        enc_c = enc.contiguous()
        kvpk = proc.project_kv_frags(attn, enc_c)
        scale_dev = proc.scale_device(x2d.device)
    else:
        e, Lr, enc_bs = _enc_rows(enc)
        ldt = _round8(L)
        k = torch.empty((B * L, inner), dtype=x2d.dtype, device=x2d.device)
        vt = torch.zeros((B, inner, ldt), dtype=x2d.dtype, device=x2d.device)
        w2, b2 = kv_weight(attn)
        ops.gemm(e, w2, B * L, 2 * inner, e.shape[2], bias=b2, rows_per_batch=L, out=k, n_split=inner, out_t=vt, ldt=ldt,
                 a_rows_per_batch=L, a_batch_stride=enc_bs)
        kvpk = ops.rc_kv_pack(k, vt, ldt, L, None, None, 0, 0, B)
        scale_dev = None
    tq = [attn.to_q.weight, norm.weight, norm.bias] + ([attn.to_q.bias] if attn.to_q.bias is not None else [])
    wq = _cached(attn, f"rc_xq_{float(attn.scale)!r}", tq, lambda: rowchain.pack_xattn_q(attn.to_q.weight, attn.to_q.bias, norm.weight, norm.bias, attn.scale))
    to = [attn.to_out[0].weight] + ([attn.to_out[0].bias] if attn.to_out[0].bias is not None else [])
    wo = _cached(attn, "rc_xo", to, lambda: rowchain.pack_xattn_out(attn.to_out[0].weight, attn.to_out[0].bias))
    return ops.rc_xattn(x2d, wq, kvpk, wo, N, norm.eps, T, ip_scale=scale_dev, text_len=L)


class AttnProcessor(nn.Module):
    """Self-attention (and plain cross-attention) processor — reference :282-393."""

    def __init__(self, hidden_size=None, cross_attention_dim=None):
        super().__init__()

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 return_attntion_probs=False, attn_key=None, attn_process_fn=None, return_cond_ca_only=False,
                 return_token_ca_only=None, offload_cross_attn_to_cpu=False, save_attn_to_dict=None, save_keys=None,
                 enable_flash_attn=True, _fused_residual=None, _fused_ln=None):
        """``_fused_ln`` = (nn.LayerNorm, row statistics or None) (internal, ``BasicTransformerBlock``): ``hidden_states`` is the block's
        UN-normalised stream and the norm is folded into the first projection (tg_gemm ``ln_u`` / ``ln_v`` / ``ln_rows``)."""
        _check_common(attn, hidden_states, attn_process_fn)
        x, B, N, C, shape4 = _to_tokens(hidden_states)
        inner, heads, d = attn_dims(attn)
        xin = x                                                 # the residual of `residual_connection` is the UN-normalised input (:313)
        if getattr(attn, "group_norm", None) is not None:
            if _fused_ln is not None:
                raise RuntimeError("theatergen_amd: group_norm and a folded LayerNorm cannot both precede the projections")
            x = _group_norm_tokens(attn.group_norm, x, B, N)
        o = torch.empty((B * N, inner), dtype=x.dtype, device=x.device)
        lnq = None
        if _fused_ln is not None:
            norm, rows = _fused_ln                           # (nn.LayerNorm, layernorm_stats tensor or None = statistics inside the kernel)
            wl, ul, vl = ln_weight(attn, "qkv" if encoder_hidden_states is None else "q", norm)
            lnq = (ul, vl, norm.eps, rows)
        if encoder_hidden_states is None:
            # one GEMM: [Q | K] token-major + V^T per batch item
            mask = _prepare_mask(attention_mask, N, B, heads, N) if attention_mask is not None else None
            ldt = _round8(N)
            qk = torch.empty((B * N, 2 * inner), dtype=x.dtype, device=x.device)
            vt = torch.empty((B, inner, ldt), dtype=x.dtype, device=x.device)
            w3, b3 = (wl, None) if lnq else qkv_weight(attn)
            ops.gemm(x, w3, B * N, 3 * inner, C, bias=b3, rows_per_batch=N, out=qk, n_split=2 * inner, out_t=vt, ldt=ldt, ln=lnq)
            ops.attention(qk, 2 * inner, N * 2 * inner, qk[:, inner:], 2 * inner, N * 2 * inner, vt, ldt, inner * ldt, N,
                          B, heads, d, N, attn.scale, o, inner, N * inner, mask=mask)
        else:
            if getattr(attn, "norm_cross", None):
                encoder_hidden_states = _norm_encoder(attn, encoder_hidden_states)
            enc, L, enc_bs = _enc_rows(encoder_hidden_states)
            mask = _prepare_mask(attention_mask, L, B, heads, N) if attention_mask is not None else None
            ctx = enc.shape[2]
            ldt = _round8(L)
            k = torch.empty((B * L, inner), dtype=x.dtype, device=x.device)
            vt = torch.empty((B, inner, ldt), dtype=x.dtype, device=x.device)
            w2, b2 = kv_weight(attn)
            ops.gemm(enc, w2, B * L, 2 * inner, ctx, bias=b2, rows_per_batch=L, out=k, n_split=inner,
                     out_t=vt, ldt=ldt, a_rows_per_batch=L, a_batch_stride=enc_bs)
            ca_kw = dict(save_attn_to_dict=save_attn_to_dict, attention_mask=attention_mask, attn_process_fn=attn_process_fn,
                         return_attntion_probs=return_attntion_probs)
            if lnq and rows is None and xq_eligible(attn, x, B, N, L, 0, ca_kw):
                # inner levels: norm2 + to_q + attention in ONE launch (q never exists)
                wx, ux, vx = xq_weight(attn, norm)
                blob = ops.xq_kv_pack(k, vt, ldt, L, None, None, 0, 0, B, inner, d)
                ops.xq_attn(x, wx, ux, vx, norm.eps, blob, d, N, L, 0, out=o)
                return _finish(attn, o, B, N, C, shape4, xin, _fused_residual)
            q = ops.linear(x, wl, ln=lnq) if lnq else ops.linear(x, attn.to_q.weight, attn.to_q.bias)
            ops.attention(q, inner, N * inner, k, inner, L * inner, vt, ldt, inner * ldt, L, B, heads, d, N, attn.scale,
                          o, inner, N * inner, mask=mask)
            # like the reference (:371) self.return_attntion_probs is forced False; maps are still saved (:386-389)
            if save_attn_to_dict is not None:
                if mask is not None:
                    raise NotImplementedError("theatergen_amd: attention-map capture with an attention_mask is not supported")
                _save_probs(attn, q, inner, k, inner, B, N, L, save_attn_to_dict, save_keys, attn_key, return_cond_ca_only,
                            return_token_ca_only, bool(save_attn_to_dict) or offload_cross_attn_to_cpu)
        return _finish(attn, o, B, N, C, shape4, xin, _fused_residual)


AttentionProcessor = AttnProcessor


class IPAttnProcessor(nn.Module):
    """Decoupled text + image cross-attention for IP-Adapter — reference :396-553.

    ``O = softmax(s Q Kt^T) Vt + scale * softmax(s Q Kip^T) Vip`` in one fused kernel; the last ``num_tokens``
    rows of ``encoder_hidden_states`` are the image tokens (:467-471)."""

    def __init__(self, hidden_size, cross_attention_dim=None, scale=1.0, num_tokens=4):
        super().__init__()
        self.hidden_size = hidden_size
        self.cross_attention_dim = cross_attention_dim
        self._scale_dev = None          # fp32 [1] on the compute device: what the kernel reads (see the ``scale`` property)
        self.scale = scale
        self.num_tokens = num_tokens
        self.to_k_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)
        self.to_v_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)
        self._packed = None
        self._kv = StaticSlots()

    # ``scale`` stays the plain mutable attribute of the reference (``IPAdapter.set_scale`` assigns it per character,
    # ip_adapter/ip_adapter.py:155-158; ``custom_pipelines.py:328-333`` toggles 0.0 <-> s per step), but the value the attention
    # kernel multiplies with lives in a DEVICE scalar: an assignment refreshes that scalar (one 4-byte fill on the current
    # stream), so a hipGraph captured with one scale replays with the next one — no re-capture, no host sync.
    # STREAM CONTRACT (round 5): the fill is issued on the current stream AND fenced against every registered replay stream (``fenced_fill``:
    # the engine-owned streams of ``DenoiseEngine.run_concurrent`` register themselves), so an assignment between two replays — on whichever stream
    # they run — is seen by the second and never lands under the first.  Engines that share one UNet share this scalar (one IP scale per UNet at a
    # time); assigning it from ANOTHER THREAD while a loop runs is still unordered on the host side.
    @property
    def scale(self):
        return self._scale

    @scale.setter
    def scale(self, value):
        self._scale = value
        if self._scale_dev is not None:
            fenced_fill(self._scale_dev, float(value))

    def scale_device(self, device):
        if self._scale_dev is None or self._scale_dev.device != device:
            self._scale_dev = torch.full((1,), float(self._scale), dtype=torch.float32, device=device)
        return self._scale_dev

    def _ip_weight(self):
        ws = (self.to_k_ip.weight, self.to_v_ip.weight)
        key = tuple((t.data_ptr(), tensor_version(t), t.dtype, t.device) for t in ws)
        if self._packed is None or self._packed[0] != key:
            with torch.no_grad():
                self._packed = (key, torch.cat([w.detach() for w in ws], dim=0).contiguous())
        return self._packed[1]

    def _weights_key(self, attn):
        return (self.num_tokens, tensor_version(attn.to_k.weight), tensor_version(attn.to_v.weight), attn.to_k.weight.data_ptr(),
                tensor_version(self.to_k_ip.weight), tensor_version(self.to_v_ip.weight), self.to_k_ip.weight.data_ptr())

    def _project_into(self, attn, enc, bufs):
        """text K | V^T and image K | V^T of ``enc`` [B, L + T, ctx] -> ``bufs`` (two small GEMMs, V written transposed)"""
        B, Ltot, ctx = enc.shape
        T = self.num_tokens
        L = Ltot - T
        inner = attn_dims(attn)[0]
        k, vt, kip, vtip = bufs
        ldt, ldi = _round8(L), _round8(T)
        w2, b2 = kv_weight(attn)
        ops.gemm(enc, w2, B * L, 2 * inner, ctx, bias=b2, rows_per_batch=L, out=k, n_split=inner, out_t=vt, ldt=ldt,
                 a_rows_per_batch=L, a_batch_stride=Ltot * ctx)
        ops.gemm(enc.reshape(-1)[L * ctx:], self._ip_weight(), B * T, 2 * inner, ctx, rows_per_batch=T, out=kip,
                 n_split=inner, out_t=vtip, ldt=ldi, a_rows_per_batch=T, a_batch_stride=Ltot * ctx)
        return (k, vt, ldt, kip, vtip, ldi, L, T)

    def _alloc(self, attn, enc):
        B, Ltot, _ = enc.shape
        T = self.num_tokens
        L = Ltot - T
        if L < 1 or T < 1 or T > 64:
            raise RuntimeError(f"IPAttnProcessor: need 1 <= num_tokens <= 64 and at least one text token (L={L}, T={T})")
        inner = attn_dims(attn)[0]
        kw = dict(dtype=enc.dtype, device=enc.device)
        # the pad columns of V^T (keys L..ldt) are multiplied by exact-zero probabilities: they must be finite
        return (torch.empty((B * L, inner), **kw), torch.zeros((B, inner, _round8(L)), **kw),
                torch.empty((B * T, inner), **kw), torch.zeros((B, inner, _round8(T)), **kw))

    def register_static(self, attn, enc):
        """Called by the OWNER of a persistent conditioning buffer (``DenoiseEngine.set_conditioning``) after it copied new
        embeddings into it: (re)project K / V^T into buffers that belong to that tensor.  The buffers are allocated once per
        (tensor, shape) and refreshed IN PLACE, so a captured hipGraph of the UNet step keeps valid pointers."""
        slot = self._kv.get(enc)
        bkey = (tuple(enc.shape), enc.dtype, enc.device, self.num_tokens, attn_dims(attn)[0])
        if slot is None or slot["bkey"] != bkey or slot["attn"]() is not attn:
            slot = self._kv.put(enc, {"bkey": bkey, "attn": weakref.ref(attn), "bufs": self._alloc(attn, enc)})
        slot["kv"] = self._project_into(attn, enc, slot["bufs"])
        slot["key"] = (tensor_version(enc), self._weights_key(attn))
        if slot.get("kvpk") is not None:
            self._pack_frags(slot["kv"], enc.shape[0], slot["kvpk"])     # refreshed IN PLACE with the projections (graph-stable pointer)
        if slot.get("xqpk") is not None:
            self._pack_xq(attn, slot["kv"], enc.shape[0], slot["xqpk"])
        return slot["kv"]

    @staticmethod
    def _pack_frags(kv, B, out=None):
        k, vt, ldt, kip, vtip, ldi, L, T = kv
        return ops.rc_kv_pack(k, vt, ldt, L, kip, vtip, ldi, T, B, out=out)

    @staticmethod
    def _pack_xq(attn, kv, B, out=None):
        k, vt, ldt, kip, vtip, ldi, L, T = kv
        inner, _, d = attn_dims(attn)
        return ops.xq_kv_pack(k, vt, ldt, L, kip, vtip, ldi, T, B, inner, d, out=out)

    def project_kv_xq(self, attn, enc, kv=None):
        """K / V^T of ``enc`` as the MFMA fragments of the fused inner-level launch (``tg_xq_attn``): for a REGISTERED tensor they live in the
        tensor's slot and are re-packed in place whenever the projections are; any other tensor is packed per call.  ``kv``: the tuple a caller
        already got from ``project_kv`` for this very call (an unregistered tensor would otherwise be projected twice)."""
        if kv is None:
            kv = self.project_kv(attn, enc)
        slot = self._kv.get(enc)
        if slot is not None and slot["attn"]() is attn and slot.get("kv") is kv:
            if slot.get("xqpk") is None:
                slot["xqpk"] = self._pack_xq(attn, kv, enc.shape[0])
            return slot["xqpk"]
        return self._pack_xq(attn, kv, enc.shape[0])

    def project_kv_frags(self, attn, enc, kv=None):
        """K / V^T of ``enc`` as the fragment blocks of the fused first-level cross-attention (``tg_rc_xattn``): for a REGISTERED
        tensor they live in the tensor's slot and are re-packed whenever the projections are; any other tensor is packed per call."""
        if kv is None:
            kv = self.project_kv(attn, enc)
        slot = self._kv.get(enc)
        if slot is not None and slot["attn"]() is attn and slot.get("kv") is kv:
            if slot.get("kvpk") is None:
                slot["kvpk"] = self._pack_frags(kv, enc.shape[0])
            return slot["kvpk"]
        return self._pack_frags(kv, enc.shape[0])

    def project_kv(self, attn, enc):
        """Text and image K / V^T of ``encoder_hidden_states``.  Step-invariant, but only a REGISTERED tensor (see
        ``register_static``) is served from its cache — and re-projected in place if it or the weights changed since; any
        other tensor (the eager drop-in path: a fresh ``torch.cat`` per character) is projected on every call."""
        slot = self._kv.get(enc)
        if slot is not None and slot["attn"]() is attn:
            ver = tensor_version(enc)
            if ver is not None and slot["key"] == (ver, self._weights_key(attn)):
                return slot["kv"]
            return self.register_static(attn, enc)
        return self._project_into(attn, enc, self._alloc(attn, enc))

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 return_attntion_probs=False, attn_key=None, attn_process_fn=None, return_cond_ca_only=False,
                 return_token_ca_only=None, offload_cross_attn_to_cpu=False, save_attn_to_dict=None, save_keys=None,
                 enable_flash_attn=True, _fused_residual=None, _fused_ln=None):
        _check_common(attn, hidden_states, attn_process_fn)
        if encoder_hidden_states is None:
            raise RuntimeError("IPAttnProcessor is a cross-attention processor: encoder_hidden_states is required")
        if attention_mask is not None:
            # reference :461-463 prepares the mask for the FULL (text + image) sequence and :477 adds it to text-only scores: baddbmm
            # rejects the shapes for every mask length, so there is no behaviour to reproduce
            raise RuntimeError("IPAttnProcessor: attention_mask cannot be combined with the decoupled text / image split "
                               "(the reference's baddbmm fails on the shapes as well)")
        x, B, N, C, shape4 = _to_tokens(hidden_states)
        inner, heads, d = attn_dims(attn)
        xin = x
        if getattr(attn, "group_norm", None) is not None:
            x = _group_norm_tokens(attn.group_norm, x, B, N)
        if getattr(attn, "norm_cross", None):
            # the reference normalises the whole [text ; image] sequence before the split (:465-466); a normalised copy is a fresh tensor,
            # so the registered-tensor K / V^T cache does not apply to it
            encoder_hidden_states = _norm_encoder(attn, encoder_hidden_states)
        enc = encoder_hidden_states.contiguous()
        kv_now = self.project_kv(attn, enc)
        k, vt, ldt, kip, vtip, ldi, L, T = kv_now
        if _fused_ln is not None and _fused_ln[1] is None and xq_eligible(attn, x, B, N, L, T, dict(
                save_attn_to_dict=save_attn_to_dict, attn_process_fn=attn_process_fn, return_attntion_probs=return_attntion_probs)):
            # inner levels (round 5): norm2 + to_q + the two softmaxes + PV in ONE launch; the K / V^T fragments live with the conditioning's slot
            norm = _fused_ln[0]
            wx, ux, vx = xq_weight(attn, norm)
            o = torch.empty((B * N, inner), dtype=x.dtype, device=x.device)
            ops.xq_attn(x, wx, ux, vx, norm.eps, self.project_kv_xq(attn, enc, kv=kv_now), d, N, L, T, ip_scale=self.scale_device(x.device), out=o)
            return _finish(attn, o, B, N, C, shape4, xin, _fused_residual)
        if _fused_ln is not None:
            norm, rows = _fused_ln
            wl, ul, vl = ln_weight(attn, "q", norm)
            q = ops.linear(x, wl, ln=(ul, vl, norm.eps, rows))
        else:
            q = ops.linear(x, attn.to_q.weight, attn.to_q.bias)
        o = torch.empty((B * N, inner), dtype=x.dtype, device=x.device)
        ops.attention(q, inner, N * inner, k, inner, L * inner, vt, ldt, inner * ldt, L, B, heads, d, N, attn.scale,
                      o, inner, N * inner, k1=kip, k1_ld=inner, k1_bs=T * inner, vt1=vtip, vt1_ld=ldi, vt1_bs=inner * ldi,
                      len1=T, w1=float(self.scale), w1_dev=self.scale_device(x.device))
        out = _finish(attn, o, B, N, C, shape4, xin, _fused_residual)
        if return_attntion_probs or save_attn_to_dict is not None:
            probs = _save_probs(attn, q, inner, k, inner, B, N, L, save_attn_to_dict, save_keys, attn_key,
                                return_cond_ca_only, return_token_ca_only, offload_cross_attn_to_cpu)
            if return_attntion_probs:
                return out, probs
        return out


class CNAttnProcessor:
    """ControlNet processor: attends to the text tokens only — reference :861-923."""

    def __init__(self, num_tokens=4):
        self.num_tokens = num_tokens
        self._plain = AttnProcessor()

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, _fused_residual=None,
                 _fused_ln=None):
        if encoder_hidden_states is not None:
            end_pos = encoder_hidden_states.shape[1] - self.num_tokens
            encoder_hidden_states = encoder_hidden_states[:, :end_pos]
        return self._plain(attn, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask,
                           temb=temb, _fused_residual=_fused_residual, _fused_ln=_fused_ln)


# the reference's torch-2 aliases (ip_adapter/ip_adapter.py:13-24 selects these names)
AttnProcessor2_0 = AttnProcessor
IPAttnProcessor2_0 = IPAttnProcessor
CNAttnProcessor2_0 = CNAttnProcessor

"""Oracle (test infrastructure): restatement of the reference attention processors.

Follows ``ip_adapter/attention_processor.py`` of the reference:
  * ``Attention`` module                          :12-279
  * ``AttnProcessor.__call__`` (self-attention)   :294-393
  * ``IPAttnProcessor.__call__`` (decoupled)      :421-553
  * ``CNAttnProcessor.__call__`` (text-only)      :868-923
Pinned against the imported reference on ``tests/golden/attn.npz`` and ``attn_branches.npz`` (tests/test_oracle_golden.py).

All functions are torch-CPU fp32, op for op in the reference's order.  Weights
come in a plain dict with the reference's parameter names:
``to_q.weight [inner,C]``, ``to_k.weight [inner,ctx]``, ``to_v.weight``,
``to_out.0.weight [C,inner]``, ``to_out.0.bias [C]`` and, for the IP processor,
``to_k_ip.weight`` / ``to_v_ip.weight`` ``[C,ctx]``.
"""
import torch
import torch.nn.functional as F


def head_to_batch_dim(t, heads):
    """reference attention_processor.py:176-185"""
    b, n, dim = t.shape
    t = t.reshape(b, n, heads, dim // heads).permute(0, 2, 1, 3)
    return t.reshape(b * heads, n, dim // heads)


def batch_to_head_dim(t, heads):
    """reference attention_processor.py:169-174"""
    bh, n, d = t.shape
    t = t.reshape(bh // heads, heads, n, d).permute(0, 2, 1, 3)
    return t.reshape(bh // heads, n, d * heads)


def get_attention_scores(q, k, scale):
    """reference attention_processor.py:187-219 (no mask, no upcast):
    ``baddbmm(empty, q, k^T, beta=0, alpha=scale)`` then ``softmax(-1)``."""
    scores = torch.baddbmm(
        torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype),
        q, k.transpose(-1, -2), beta=0, alpha=scale)
    return scores.softmax(dim=-1)


def _select_probs(probs, batch, heads, return_token_ca_only, return_cond_ca_only):
    """reference attention_processor.py:532-545 (attention-map capture)."""
    p = probs.unflatten(0, (batch, heads))
    if return_token_ca_only is not None:
        if isinstance(return_token_ca_only, int):
            p = p[:, :, :, return_token_ca_only:return_token_ca_only + 1]
        else:
            p = p[:, :, :, return_token_ca_only]
    if return_cond_ca_only:
        assert batch % 2 == 0
        p = p[batch // 2:]
    return p


def _pre(x):
    ndim = x.ndim
    shape4 = None
    if ndim == 4:
        b, c, h, w = x.shape
        shape4 = (b, c, h, w)
        x = x.view(b, c, h * w).transpose(1, 2)
    return x, shape4


def _post(w, x, shape4, residual, residual_connection, rescale_output_factor):
    x = F.linear(x, w["to_out.0.weight"], w.get("to_out.0.bias"))
    if shape4 is not None:
        b, c, h, wd = shape4
        x = x.transpose(-1, -2).reshape(b, c, h, wd)
    if residual_connection:
        x = x + residual
    return x / rescale_output_factor


def prepare_attention_mask(mask, target_length, batch, heads):
    """reference attention_processor.py:221-259 (out_dim = 3)"""
    if mask is None:
        return None
    if mask.shape[-1] != target_length:
        mask = F.pad(mask, (0, target_length), value=0.0)
    if mask.shape[0] < batch * heads:
        mask = mask.repeat_interleave(heads, dim=0)
    return mask


def norm_encoder_hidden_states(w, enc, kind, groups=32):
    """reference attention_processor.py:261-279: LayerNorm(ctx) or GroupNorm over [B, ctx, L]"""
    if kind == "layer_norm":
        return F.layer_norm(enc, (enc.shape[-1],), w["norm_cross.weight"], w["norm_cross.bias"], 1e-5)
    if kind == "group_norm":
        return F.group_norm(enc.transpose(1, 2), groups, w["norm_cross.weight"], w["norm_cross.bias"], 1e-5).transpose(1, 2)
    raise ValueError(kind)


def attn_processor(w, heads, hidden_states, encoder_hidden_states=None,
                   residual_connection=False, rescale_output_factor=1.0,
                   return_probs=False, return_token_ca_only=None, return_cond_ca_only=False,
                   attention_mask=None, norm_num_groups=None, eps=1e-5, cross_attention_norm=None,
                   cross_attention_norm_num_groups=32):
    """``AttnProcessor.__call__`` — reference attention_processor.py:294-393, including the pre-projection branches (:319-341):
    ``prepare_attention_mask``, ``group_norm`` on the [B, C, N] view, ``norm_cross`` on the encoder states, q/k/v bias
    (``to_q.bias`` ... in ``w``) and the additive mask of ``baddbmm(mask, q, k^T, beta=1, alpha=scale)`` (:193-206).
    Returns ``out`` or ``(out, probs)`` (the tensor the reference stores in ``save_attn_to_dict``)."""
    residual = hidden_states
    x, shape4 = _pre(hidden_states)
    batch = x.shape[0]
    seq = x.shape[1] if encoder_hidden_states is None else encoder_hidden_states.shape[1]
    mask = prepare_attention_mask(attention_mask, seq, batch, heads)
    if norm_num_groups is not None:
        x = F.group_norm(x.transpose(1, 2), norm_num_groups, w["group_norm.weight"], w["group_norm.bias"], eps).transpose(1, 2)
    q = F.linear(x, w["to_q.weight"], w.get("to_q.bias"))
    if encoder_hidden_states is None:
        enc = x
    elif cross_attention_norm:
        enc = norm_encoder_hidden_states(w, encoder_hidden_states, cross_attention_norm, cross_attention_norm_num_groups)
    else:
        enc = encoder_hidden_states
    k = F.linear(enc, w["to_k.weight"], w.get("to_k.bias"))
    v = F.linear(enc, w["to_v.weight"], w.get("to_v.bias"))
    d = q.shape[-1] // heads
    scale = d ** -0.5
    q, k, v = (head_to_batch_dim(t, heads) for t in (q, k, v))
    if mask is None:
        probs = get_attention_scores(q, k, scale)
    else:
        probs = torch.baddbmm(mask, q, k.transpose(-1, -2), beta=1, alpha=scale).softmax(dim=-1)
    o = batch_to_head_dim(torch.bmm(probs, v), heads)
    out = _post(w, o, shape4, residual, residual_connection, rescale_output_factor)
    if return_probs:
        return out, _select_probs(probs, batch, heads, return_token_ca_only, return_cond_ca_only)
    return out


def ip_attn_processor(w, heads, hidden_states, encoder_hidden_states, scale_ip, num_tokens,
                      residual_connection=False, rescale_output_factor=1.0,
                      return_probs=False, return_token_ca_only=None, return_cond_ca_only=False):
    """``IPAttnProcessor.__call__`` — reference attention_processor.py:421-553.
    ``O = softmax(s Q Kt^T) Vt + scale_ip * softmax(s Q Kip^T) Vip`` with TWO
    independent softmaxes (:482, :503, :516); only the TEXT probabilities are
    exported (:532)."""
    residual = hidden_states
    x, shape4 = _pre(hidden_states)
    batch = x.shape[0]
    q = F.linear(x, w["to_q.weight"])
    end_pos = encoder_hidden_states.shape[1] - num_tokens
    enc, ip = encoder_hidden_states[:, :end_pos, :], encoder_hidden_states[:, end_pos:, :]
    k = F.linear(enc, w["to_k.weight"])
    v = F.linear(enc, w["to_v.weight"])
    d = q.shape[-1] // heads
    scale = d ** -0.5
    q, k, v = (head_to_batch_dim(t, heads) for t in (q, k, v))
    probs = get_attention_scores(q, k, scale)
    o = batch_to_head_dim(torch.bmm(probs, v), heads)
    ip_k = head_to_batch_dim(F.linear(ip, w["to_k_ip.weight"]), heads)
    ip_v = head_to_batch_dim(F.linear(ip, w["to_v_ip.weight"]), heads)
    ip_probs = get_attention_scores(q, ip_k, scale)
    ip_o = batch_to_head_dim(torch.bmm(ip_probs, ip_v), heads)
    o = o + scale_ip * ip_o
    out = _post(w, o, shape4, residual, residual_connection, rescale_output_factor)
    if return_probs:
        return out, _select_probs(probs, batch, heads, return_token_ca_only, return_cond_ca_only)
    return out


def cn_attn_processor(w, heads, hidden_states, encoder_hidden_states=None, num_tokens=4,
                      residual_connection=False, rescale_output_factor=1.0):
    """``CNAttnProcessor.__call__`` — reference attention_processor.py:868-923:
    the ControlNet branch attends to the text tokens only (:894-895)."""
    if encoder_hidden_states is not None:
        end_pos = encoder_hidden_states.shape[1] - num_tokens
        encoder_hidden_states = encoder_hidden_states[:, :end_pos]
    return attn_processor(w, heads, hidden_states, encoder_hidden_states,
                          residual_connection, rescale_output_factor)

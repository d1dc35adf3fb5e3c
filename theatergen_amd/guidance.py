"""Per-box cross-attention guidance (reference ``utils/guidance.py``), device side in HIP.

Same call surface as the reference: ``compute_ca_lossv3(saved_attn, bboxes, object_positions,
guidance_attn_keys, **kwargs)`` (:244-286) and ``get_phrase_indices`` (:32-89, host string work).  The loss
reductions (masked top-k means / ratio) run in ``tg_guidance_topk`` / ``tg_guidance_ratio``; because the HIP
path has no autograd, ``return_grads=True`` additionally returns the analytic d loss / d attention-map per key
(what ``latent_backward_guidance``, reference ``models/pipelines.py:62-128``, back-propagates from).
Box -> mask rounding is host integer work shared with ``theatergen_amd.utils.scale_proportion``.
"""
import math
from collections.abc import Iterable

import torch

from . import ops
from .utils import scale_proportion


_mask_cache = {}           # (boxes, H, W, device) -> (host mask, device mask): box masks are step-invariant, uploaded once
_MASK_CACHE_MAX = 256
_mask_pinned = {}          # masks a stream capture has read: their device address is baked into a hipGraph, so they are never evicted


def _box_mask(obj_boxes, H, W, device):
    if not isinstance(obj_boxes[0], Iterable):
        obj_boxes = [obj_boxes]
    key = (tuple(tuple(float(v) for v in bx) for bx in obj_boxes), int(H), int(W), str(device))
    capturing = ops.capturing_now()
    hit = _mask_pinned.get(key) or _mask_cache.get(key)
    if hit is None:
        if capturing:
            raise RuntimeError("guidance: this box mask is not on the device yet and a stream capture is in progress (the upload cannot "
                               "be captured): run the same call once eagerly before capturing it")
        m = torch.zeros(H, W)
        for bx in obj_boxes:
            x0, y0, x1, y1 = scale_proportion(bx, H=H, W=W)
            m[y0:y1, x0:x1] = 1
        if len(_mask_cache) >= _MASK_CACHE_MAX:
            _mask_cache.pop(next(iter(_mask_cache)))
        hit = _mask_cache[key] = (m, m.to(device))
    if capturing:
        _mask_pinned[key] = hit
    return hit


def add_ca_loss_per_attn_map_to_loss(loss, attn_map, object_number, bboxes, object_positions, use_ratio_based_loss=True,
                                     fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=1.0, verbose=False,
                                     grad=None, scale=1.0, batch=None):
    """reference guidance.py:91-148.  ``loss``: fp32 device scalar tensor [1] accumulated IN PLACE;
    ``attn_map`` fp32 [heads, HW, tokens]; ``grad`` (optional) same shape, accumulated in place.  With ``batch`` (an
    ``ops.GuidanceBatch``) the terms are only queued; the caller flushes them in one launch."""
    b, i, j = attn_map.shape
    H = W = int(math.sqrt(i))
    attn_map = attn_map.contiguous()
    for obj_idx in range(object_number):
        mask_host, mask = _box_mask(bboxes[obj_idx], H, W, attn_map.device)
        n_pos = len(object_positions[obj_idx])
        if not use_ratio_based_loss:
            # k = max(1, floor(p * count)) computed like the reference: fp32 product then .long() (:136-137)
            msum = mask_host.sum()
            k_fg = int((msum * fg_top_p).long().clamp_(min=1))
            k_bg = int(((1 - mask_host).sum() * bg_top_p).long().clamp_(min=1))
        for pos in object_positions[obj_idx]:
            if batch is not None:
                if use_ratio_based_loss:
                    batch.add(batch.KIND_RATIO, attn_map, pos, mask, scale / n_pos, grad)
                else:
                    batch.add(batch.KIND_TOPK, attn_map, pos, mask, scale / n_pos, grad, k_fg=k_fg, k_bg=k_bg, fg_w=fg_weight, bg_w=bg_weight)
            elif use_ratio_based_loss:
                ops.guidance_ratio(attn_map, pos, mask, scale / n_pos, loss, grad)
            else:
                ops.guidance_topk(attn_map, pos, mask, k_fg, k_bg, fg_weight, bg_weight, scale / n_pos, loss, grad)
    return loss


def add_ref_ca_loss_per_attn_map_to_lossv2(loss, saved_attn, object_number, bboxes, object_positions, guidance_attn_keys,
                                           ref_ca_saved_attns, ref_ca_last_token_only, ref_ca_word_token_only,
                                           word_token_indices, index, loss_weight, eps=1e-5, verbose=False, grads=None,
                                           scale=1.0, batch=None):
    """reference guidance.py:150-242 (attention transfer from saved per-box reference maps).  ``loss``: fp32 device
    scalar tensor [1] accumulated IN PLACE; ``grads``: optional {key: fp32 tensor like saved_attn[key]} accumulated in
    place; ``scale`` multiplies every added term (compute_ca_lossv3's 1 / (n_obj * n_keys))."""
    if loss_weight == 0.0:                                    # :156-157
        return loss
    for obj_idx in range(object_number):
        obj_boxes = bboxes[obj_idx]
        refs = ref_ca_saved_attns[obj_idx]
        if not isinstance(obj_boxes[0], Iterable):            # :163-166
            obj_boxes, refs = [obj_boxes], [refs]
        assert len(obj_boxes) == len(refs), f"{len(obj_boxes)} != {len(refs)}"
        if ref_ca_word_token_only:                            # :207-217
            positions = [word_token_indices[obj_idx]]
        elif ref_ca_last_token_only:
            positions = [object_positions[obj_idx][-1]]
        else:
            positions = object_positions[obj_idx]
        term = scale * loss_weight / (len(obj_boxes) * len(positions))       # :237
        for bx, ref in zip(obj_boxes, refs):
            ref = ref[index]
            for key in guidance_attn_keys:
                amap = saved_attn[key]
                amap3 = (amap.squeeze(dim=0) if amap.dim() == 4 else amap).contiguous()
                b, i, _ = amap3.shape
                H = W = int(math.sqrt(i))
                rmap = ref[key]
                assert rmap.ndim == 4                         # [1, heads, HW, 1]  (:184-186)
                rcol = rmap[0, :, :, 0].to(device=amap3.device, dtype=torch.float32).contiguous()
                _, mask = _box_mask(bx, H, W, amap3.device)
                g = None
                if grads is not None:
                    g = grads[key]
                    g = g.squeeze(dim=0) if g.dim() == 4 else g
                for pos in positions:
                    if batch is not None:
                        if rcol.numel() != amap3.shape[0] * amap3.shape[1]:
                            raise RuntimeError("guidance: reference map shape does not match the attention map")
                        batch.add(batch.KIND_REF, amap3, pos, mask, term, g, ref=rcol, eps=eps)
                    else:
                        ops.guidance_ref(amap3, pos, rcol, mask, eps, term, loss, g)
    return loss


def compute_ca_lossv3(saved_attn, bboxes, object_positions, guidance_attn_keys, ref_ca_saved_attns=None,
                      ref_ca_last_token_only=True, ref_ca_word_token_only=False, word_token_indices=None, index=None,
                      ref_ca_loss_weight=1.0, verbose=False, return_grads=False, loss_scale=1.0, **kwargs):
    """reference guidance.py:244-286, including the reference-attention transfer term (:150-242, :278-284) when
    ``ref_ca_saved_attns`` is given.  ``loss_scale`` (extension): every term — value and gradient — is multiplied by it on the
    device (``latent_backward_guidance`` differentiates ``loss * loss_scale``, models/pipelines.py:94)."""
    object_number = len(bboxes)
    dev = None
    for k in guidance_attn_keys:
        dev = saved_attn[k].device
        break
    loss = torch.zeros(1, dtype=torch.float32, device=dev if dev is not None else "cuda")
    grads = {}
    if object_number == 0 or len(guidance_attn_keys) == 0:
        return (loss[0], grads) if return_grads else loss[0]
    norm = float(loss_scale) / (object_number * len(guidance_attn_keys))
    batch = ops.GuidanceBatch(loss.device)          # every term of this call: one launch + an in-order fold
    for key in guidance_attn_keys:
        amap = saved_attn[key]
        if amap.dtype != torch.float32:
            raise RuntimeError("guidance attention maps must be fp32 (tg_attn_probs exports fp32)")
        amap3 = amap.squeeze(dim=0) if amap.dim() == 4 else amap
        g = None
        if return_grads:
            g = torch.zeros_like(amap3)
            grads[key] = g.reshape(amap.shape)
        add_ca_loss_per_attn_map_to_loss(loss, amap3.contiguous(), object_number, bboxes, object_positions, grad=g, scale=norm,
                                         batch=batch, **kwargs)
    if ref_ca_saved_attns is not None:
        add_ref_ca_loss_per_attn_map_to_lossv2(loss, saved_attn, object_number, bboxes, object_positions, guidance_attn_keys,
                                               ref_ca_saved_attns, ref_ca_last_token_only, ref_ca_word_token_only,
                                               word_token_indices, index, ref_ca_loss_weight,
                                               grads=grads if return_grads else None, scale=norm, batch=batch)
    batch.flush(loss)
    return (loss[0], grads) if return_grads else loss[0]


# ---- host string work (reference guidance.py:10-89) ---------------------------------------------

def get_token_map(tokenizer, prompt, verbose=False, padding="do_not_pad"):
    ids = tokenizer([prompt], padding=padding, max_length=77, return_tensors="np")["input_ids"][0]
    return [tokenizer._convert_id_to_token(i) for i in ids.tolist()]


def get_phrase_indices(tokenizer, prompt, phrases, verbose=False, words=None, include_eos=False, token_map=None,
                       return_word_token_indices=False, add_suffix_if_not_found=False):
    for obj in phrases:
        if obj not in prompt:           # suffix missing phrases: "prompt| phrase" (:35-37)
            prompt += "| " + obj
    if token_map is None:
        token_map = get_token_map(tokenizer, prompt)
    joined = " ".join(token_map)
    object_positions, word_token_indices = [], []
    for obj_ind, obj in enumerate(phrases):
        toks = get_token_map(tokenizer, obj)[1:-1]          # strip <bos>/<eos>
        sub = " ".join(toks)
        first = len(joined[:joined.index(sub) - 1].split(" "))
        positions = list(range(first, first + len(toks)))
        if include_eos:
            positions.append(token_map.index(tokenizer.eos_token))
        object_positions.append(positions)
        if return_word_token_indices:
            if words is None:
                word_token_indices.append(object_positions[0][-1])
            else:
                wt = get_token_map(tokenizer, words[obj_ind])
                word_token_indices.append(first + toks.index(wt[-2]))
    ret = [object_positions]
    if return_word_token_indices:
        ret.append(word_token_indices)
    if add_suffix_if_not_found:
        ret.append(prompt)
    return ret[0] if len(ret) == 1 else tuple(ret)

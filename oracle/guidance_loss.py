"""Oracle (test infrastructure): per-box cross-attention-map guidance losses.

Restates reference ``utils/guidance.py``:
  * ``add_ca_loss_per_attn_map_to_loss`` :91-148 (max-based :130-144, ratio-based :122-128)
  * ``add_ref_ca_loss_per_attn_map_to_lossv2`` :150-242
  * ``compute_ca_lossv3`` :244-286
  * ``get_phrase_indices`` / ``get_token_map`` :10-89 (host string work)
Pinned against the imported reference on ``tests/golden/guidance.npz`` (values and d loss / d A).
Device-agnostic torch (the reference hard-codes ``device="cuda"``, :104, :253).
"""
import math
from collections.abc import Iterable

import torch

from .box_geometry import scale_proportion


def _box_mask(obj_boxes, H, W):
    if not isinstance(obj_boxes[0], Iterable):
        obj_boxes = [obj_boxes]
    m = torch.zeros(H, W)
    for bx in obj_boxes:
        x0, y0, x1, y1 = scale_proportion(bx, H=H, W=W)
        m[y0:y1, x0:x1] = 1
    return m


def ca_loss_one_map(attn_map, bboxes, object_positions, use_ratio_based_loss=True,
                    fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=1.0):
    """guidance.py:91-148 for ONE attention map ``[heads, HW, tokens]``; returns the added loss."""
    b, i, _ = attn_map.shape
    H = W = int(math.sqrt(i))
    loss = attn_map.new_zeros(())
    for obj_idx in range(len(bboxes)):
        mask = _box_mask(bboxes[obj_idx], H, W)
        obj_loss = 0
        for pos in object_positions[obj_idx]:
            if use_ratio_based_loss:
                ca = attn_map[:, :, pos].reshape(b, H, W)
                act = (ca * mask).reshape(b, -1).sum(dim=-1) / ca.reshape(b, -1).sum(dim=-1)
                obj_loss = obj_loss + torch.mean((1 - act) ** 2)
            else:
                ca = attn_map[:, :, pos]
                k_fg = (mask.sum() * fg_top_p).long().clamp_(min=1)
                k_bg = ((1 - mask).sum() * bg_top_p).long().clamp_(min=1)
                m1 = mask.view(1, -1)
                obj_loss = obj_loss + (1 - (ca * m1).topk(k=k_fg).values.mean(dim=1)).sum(dim=0) * fg_weight
                obj_loss = obj_loss + ((ca * (1 - m1)).topk(k=k_bg).values.mean(dim=1)).sum(dim=0) * bg_weight
        loss = loss + obj_loss / len(object_positions[obj_idx])
    return loss


def ref_ca_loss(saved_attn, bboxes, object_positions, guidance_attn_keys, ref_ca_saved_attns,
                ref_ca_last_token_only, ref_ca_word_token_only, word_token_indices, index,
                loss_weight, eps=1e-5):
    """guidance.py:150-242"""
    loss = torch.zeros(())
    if loss_weight == 0.0:
        return loss
    for obj_idx in range(len(bboxes)):
        obj_boxes = bboxes[obj_idx]
        refs = ref_ca_saved_attns[obj_idx]
        if not isinstance(obj_boxes[0], Iterable):
            obj_boxes, refs = [obj_boxes], [refs]
        assert len(obj_boxes) == len(refs)
        obj_loss = 0
        for bx, ref in zip(obj_boxes, refs):
            ref = ref[index]
            for key in guidance_attn_keys:
                amap = saved_attn[key].squeeze(dim=0)
                b, i, _ = amap.shape
                H = W = int(math.sqrt(i))
                rmap = ref[key]
                assert rmap.ndim == 4
                rmap = rmap[0, :, :, 0]
                x0, y0, x1, y1 = scale_proportion(bx, H=H, W=W)
                m = torch.zeros(H, W)
                m[y0:y1, x0:x1] = 1
                m = m.reshape(1, -1)
                if ref_ca_word_token_only:
                    positions = [word_token_indices[obj_idx]]
                elif ref_ca_last_token_only:
                    positions = [object_positions[obj_idx][-1]]
                else:
                    positions = object_positions[obj_idx]
                for pos in positions:
                    cm = amap[:, :, pos] * m
                    cm = cm / (cm.sum(dim=-1, keepdim=True) + eps)
                    rm = rmap * m
                    rm = rm / (rm.sum(dim=-1, keepdim=True) + eps)
                    obj_loss = obj_loss + torch.mean(torch.abs(cm - rm).sum(dim=-1), dim=0)
        loss = loss + loss_weight * obj_loss / (len(obj_boxes) * len(positions))
    return loss


def compute_ca_lossv3(saved_attn, bboxes, object_positions, guidance_attn_keys, ref_ca_saved_attns=None,
                      ref_ca_last_token_only=True, ref_ca_word_token_only=False, word_token_indices=None,
                      index=None, ref_ca_loss_weight=1.0, **kwargs):
    """guidance.py:244-286"""
    loss = torch.tensor(0).float()
    n_obj = len(bboxes)
    if n_obj == 0:
        return loss
    for key in guidance_attn_keys:
        loss = loss + ca_loss_one_map(saved_attn[key].squeeze(dim=0), bboxes, object_positions, **kwargs)
    n_attn = len(guidance_attn_keys)
    if n_attn > 0:
        loss = loss / (n_obj * n_attn)
    if ref_ca_saved_attns is not None:
        r = ref_ca_loss(saved_attn, bboxes, object_positions, guidance_attn_keys, ref_ca_saved_attns,
                        ref_ca_last_token_only, ref_ca_word_token_only, word_token_indices, index,
                        ref_ca_loss_weight)
        loss = loss + r / (n_obj * n_attn)
    return loss


# ---- host string work ------------------------------------------------------------------------

def get_token_map(tokenizer, prompt, padding="do_not_pad"):
    """guidance.py:10-30"""
    ids = tokenizer([prompt], padding=padding, max_length=77, return_tensors="np")["input_ids"][0]
    return [tokenizer._convert_id_to_token(i) for i in ids.tolist()]


def get_phrase_indices(tokenizer, prompt, phrases, words=None, include_eos=False, token_map=None,
                       return_word_token_indices=False, add_suffix_if_not_found=False):
    """guidance.py:32-89"""
    for obj in phrases:
        if obj not in prompt:
            prompt += "| " + obj
    if token_map is None:
        token_map = get_token_map(tokenizer, prompt)
    tm_str = " ".join(token_map)
    object_positions, word_token_indices = [], []
    for obj_ind, obj in enumerate(phrases):
        ptm = get_token_map(tokenizer, obj)[1:-1]
        ptm_str = " ".join(ptm)
        first = len(tm_str[:tm_str.index(ptm_str) - 1].split(" "))
        pos = list(range(first, first + len(ptm)))
        if include_eos:
            pos.append(token_map.index(tokenizer.eos_token))
        object_positions.append(pos)
        if return_word_token_indices:
            if words is None:
                so = object_positions[0][-1]
            else:
                wtm = get_token_map(tokenizer, words[obj_ind])
                so = first + ptm.index(wtm[-2])
            word_token_indices.append(so)
    out = (object_positions,)
    if return_word_token_indices:
        out += (word_token_indices,)
    if add_suffix_if_not_found:
        out += (prompt,)
    return out[0] if len(out) == 1 else out

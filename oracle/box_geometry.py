"""Oracle (test infrastructure): box <-> mask geometry and tensor shifting.

Restates reference ``utils/utils.py``: ``get_centered_box`` :17-42, ``proportion_to_mask`` :45-53,
``scale_proportion`` :55-68, ``binary_mask_to_box`` :70-86, ``binary_mask_to_box_mask`` :88-98,
``binary_mask_to_center`` :100-119, ``shift_tensor`` :143-178; and ``utils/schedule.py``
``get_fast_schedule`` :4-8.  Pinned against the imported reference on ``tests/golden/geometry.npz``.
Integer work here is bit-exact by construction (Python ``round`` = banker's rounding, as the reference).
"""
import numpy as np
import torch


def scale_proportion(box, H, W, use_legacy=False):
    """utils.py:55-68 — width/height are rounded separately from the corner."""
    if use_legacy:
        return int(box[0] * W), int(box[1] * H), int(box[2] * W), int(box[3] * H)
    x0, y0 = round(box[0] * W), round(box[1] * H)
    bw, bh = round((box[2] - box[0]) * W), round((box[3] - box[1]) * H)
    x1, y1 = x0 + bw, y0 + bh
    return max(x0, 0), max(y0, 0), min(x1, W), min(y1, H)


def proportion_to_mask(box, H, W, use_legacy=False):
    """utils.py:45-53"""
    x0, y0, x1, y1 = scale_proportion(box, H, W, use_legacy)
    m = torch.zeros(H, W)
    m[y0:y1, x0:x1] = 1.0
    return m


def get_centered_box(box, horizontal_center_only=True, vertical_placement="centered",
                     vertical_center=0.5, floor_padding=None):
    """utils.py:17-42"""
    x0, y0, x1, y1 = box
    w = x1 - x0
    nx0, nx1 = 0.5 - w / 2, 0.5 + w / 2
    if horizontal_center_only:
        return [nx0, y0, nx1, y1]
    h = y1 - y0
    if vertical_placement == "centered":
        assert floor_padding is None
        ny0, ny1 = vertical_center - h / 2, vertical_center + h / 2
    elif vertical_placement == "floor_padding":
        ny1 = 1 - floor_padding
        ny0 = ny1 - h
    else:
        raise ValueError(vertical_placement)
    return [nx0, ny0, nx1, ny1]


def binary_mask_to_box(mask, enlarge_box_by_one=True):
    """utils.py:70-86 (w_scale = h_scale = 1)"""
    m = np.asarray(mask).astype(bool)
    ys, xs = np.where(m)
    H, W = m.shape
    if enlarge_box_by_one:
        y0, y1 = max(int(ys.min()) - 1, 0), min(int(ys.max()) + 1, H)
        x0, x1 = max(int(xs.min()) - 1, 0), min(int(xs.max()) + 1, W)
    else:
        y0, y1, x0, x1 = int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max())
    return [x0, y0, x1, y1]


def binary_mask_to_box_mask(mask):
    """utils.py:88-98 — note the INCLUSIVE upper bound (+1)."""
    x0, y0, x1, y1 = binary_mask_to_box(mask)
    H, W = mask.shape
    out = torch.zeros(H, W)
    out[y0:y1 + 1, x0:x1 + 1] = 1.0
    return out


def binary_mask_to_center(mask, normalize=False):
    """utils.py:100-119 — centre of mass."""
    # the reference's torch branch only works for bool/integer masks (int64 matmul with arange),
    # then a true division -> float32 scalar -> .item()
    m = torch.as_tensor(np.asarray(mask)).to(torch.int64)
    h, w = m.shape
    total = m.sum()
    x = ((m.sum(dim=0) @ torch.arange(w)) / total).item()
    y = ((m.sum(dim=1) @ torch.arange(h)) / total).item()
    if normalize:
        x, y = x / w, y / h
    return x, y


def quantize_offset(x_offset, y_offset, h, w, base_w=8, base_h=8):
    """utils.py:150-153 — normalised offset -> whole multiples of 1/8 of the tensor."""
    assert h % base_h == 0 and w % base_w == 0
    return round(x_offset * base_w) * (w // base_w), round(y_offset * base_h) * (h // base_h)


def shift_tensor(t, x_offset, y_offset, offset_normalized=False, base_w=8, base_h=8):
    """utils.py:143-178 — zero-filled shift over the last two dims (ignore_last_dim=False)."""
    h, w = t.shape[-2:]
    if offset_normalized:
        x_offset, y_offset = quantize_offset(x_offset, y_offset, h, w, base_w, base_h)
    out = torch.zeros_like(t)
    ow, oh = w - abs(x_offset), h - abs(y_offset)
    ys, yd = (0, y_offset) if y_offset >= 0 else (-y_offset, 0)
    xs, xd = (0, x_offset) if x_offset >= 0 else (-x_offset, 0)
    if ow > 0 and oh > 0:
        out[..., yd:yd + oh, xd:xd + ow] = t[..., ys:ys + oh, xs:xs + ow]
    return out


def get_fast_schedule(timesteps, fast_after_steps, fast_rate):
    """utils/schedule.py:4-8"""
    if fast_after_steps >= len(timesteps) - 1:
        return timesteps
    return torch.cat((timesteps[:fast_after_steps], timesteps[fast_after_steps + 1::fast_rate]), dim=0)

"""Box <-> mask geometry (host integer work) and device-side shifting.

Mirrors reference ``utils/utils.py``: ``get_centered_box`` :17-42, ``proportion_to_mask`` :45-53,
``scale_proportion`` :55-68, ``binary_mask_to_box`` :70-86, ``binary_mask_to_box_mask`` :88-98,
``binary_mask_to_center`` :100-119, ``expand_overall_bboxes`` :135-141, ``shift_tensor`` :143-178.
Rounding is Python ``round`` (banker's), like the reference.  Masks are small (64x64): index arithmetic stays
on the host exactly as in the reference; bulk tensor movement (``shift_tensor`` of [51,1,4,64,64] latents)
runs in ``tg_shift`` on the GPU.
"""
import numpy as np
import torch

from . import ops

torch_device = "cuda"


def scale_proportion(obj_box, H, W, use_legacy=False):
    if use_legacy:
        return int(obj_box[0] * W), int(obj_box[1] * H), int(obj_box[2] * W), int(obj_box[3] * H)
    x_min, y_min = round(obj_box[0] * W), round(obj_box[1] * H)
    box_w, box_h = round((obj_box[2] - obj_box[0]) * W), round((obj_box[3] - obj_box[1]) * H)
    return max(x_min, 0), max(y_min, 0), min(x_min + box_w, W), min(y_min + box_h, H)


def proportion_to_mask(obj_box, H, W, use_legacy=False, return_np=False, device=None):
    x0, y0, x1, y1 = scale_proportion(obj_box, H, W, use_legacy)
    mask = np.zeros((H, W)) if return_np else torch.zeros(H, W)
    mask[y0:y1, x0:x1] = 1.0
    if return_np:
        return mask
    return mask.to(device if device is not None else torch_device)


def get_centered_box(box, horizontal_center_only=True, vertical_placement="centered", vertical_center=0.5, floor_padding=None):
    x_min, y_min, x_max, y_max = box
    half_w = (x_max - x_min) / 2
    out = [0.5 - half_w, y_min, 0.5 + half_w, y_max]
    if horizontal_center_only:
        return out
    h = y_max - y_min
    if vertical_placement == "centered":
        assert floor_padding is None, "Set vertical_placement to floor_padding to use floor padding"
        out[1], out[3] = vertical_center - h / 2, vertical_center + h / 2
    elif vertical_placement == "floor_padding":
        out[3] = 1 - floor_padding
        out[1] = out[3] - h
    else:
        raise ValueError(f"Unknown vertical placement: {vertical_placement}")
    return out


def binary_mask_to_box(mask, enlarge_box_by_one=True, w_scale=1, h_scale=1):
    m = mask.detach().cpu().numpy() if isinstance(mask, torch.Tensor) else np.asarray(mask)
    ys, xs = np.where(m.astype(bool))
    if ys.size == 0:
        raise ValueError("The mask is empty")
    height, width = m.shape
    if enlarge_box_by_one:
        ymin, ymax = max(int(ys.min()) - 1, 0), min(int(ys.max()) + 1, height)
        xmin, xmax = max(int(xs.min()) - 1, 0), min(int(xs.max()) + 1, width)
    else:
        ymin, ymax, xmin, xmax = int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max())
    return [xmin * w_scale, ymin * h_scale, xmax * w_scale, ymax * h_scale]


def binary_mask_to_box_mask(mask, to_device=True):
    x0, y0, x1, y1 = binary_mask_to_box(mask)
    H, W = mask.shape
    out = torch.zeros(H, W)
    out[y0:y1 + 1, x0:x1 + 1] = 1.0          # inclusive upper bound, as the reference (:96)
    return out.to(torch_device) if to_device else out


def binary_mask_to_center(mask, normalize=False):
    m = (mask.detach().cpu() if isinstance(mask, torch.Tensor) else torch.as_tensor(np.asarray(mask))).to(torch.int64)
    h, w = m.shape
    total = m.sum()
    x = ((m.sum(dim=0) @ torch.arange(w)) / total).item()
    y = ((m.sum(dim=1) @ torch.arange(h)) / total).item()
    if normalize:
        x, y = x / w, y / h
    return x, y


def expand_overall_bboxes(overall_bboxes):
    return sum(overall_bboxes, start=[])


def quantize_offset(x_offset, y_offset, h, w, base_w=8, base_h=8):
    assert h % base_h == 0 and w % base_w == 0, f"{h, w} is not a multiple of {base_h, base_w}"
    return round(x_offset * base_w) * (w // base_w), round(y_offset * base_h) * (h // base_h)


def shift_tensor(tensor, x_offset, y_offset, base_w=8, base_h=8, offset_normalized=False, ignore_last_dim=False):
    """Zero-filled shift of the last two dims.  fp32 CUDA tensors go through ``tg_shift``; small host masks
    (bool / int, 64x64) are index-copied on the host like the reference does."""
    if ignore_last_dim:
        # cross-attention maps [..., H, W, tokens] (reference utils/utils.py:146-147, 174-176): the two dims in front of the last one
        # are the image; pure data movement: shift a view with the token dim moved to the front, hand back the original layout
        moved = shift_tensor(tensor.movedim(-1, 0), x_offset, y_offset, base_w, base_h, offset_normalized, False)
        return moved.movedim(0, -1).contiguous()
    h, w = tensor.shape[-2:]
    if offset_normalized:
        x_offset, y_offset = quantize_offset(x_offset, y_offset, h, w, base_w, base_h)
    if abs(x_offset) > w or abs(y_offset) > h:
        raise RuntimeError(f"shift ({x_offset}, {y_offset}) larger than the tensor ({w}, {h})")   # the reference raises too
    if tensor.is_cuda and tensor.dtype == torch.float32:
        return ops.shift(tensor.contiguous(), x_offset, y_offset)
    out = torch.zeros_like(tensor)
    ow, oh = w - abs(x_offset), h - abs(y_offset)
    ys, yd = (0, y_offset) if y_offset >= 0 else (-y_offset, 0)
    xs, xd = (0, x_offset) if x_offset >= 0 else (-x_offset, 0)
    out[..., yd:yd + oh, xd:xd + ow] = tensor[..., ys:ys + oh, xs:xs + ow]
    return out

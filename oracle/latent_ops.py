"""Oracle (test infrastructure): seeded latents, fg/bg blending, alignment and masked composition.

Restates reference ``utils/latents.py``: ``get_unscaled_latents`` :138-149, ``blend_latents`` :156-166,
``compose_latents`` :168-218, ``align_with_bboxes`` :220-240, ``get_input_latents_list`` :257-295,
``get_input_latents_lne`` :298-325.  Pinned against the imported reference on
``tests/golden/latents.npz``.  RNG parity: noise comes from the CPU generator seeded with
``torch.manual_seed`` (:144-147, :263, :284) and EVERY object gets the same ``fg_seed_start`` (:282-283).
"""
import numpy as np
import torch

from . import box_geometry as geo


def get_unscaled_latents(batch, in_channels, height, width, seed, dtype=torch.float32):
    gen = torch.manual_seed(seed)
    return torch.randn((batch, in_channels, height // 8, width // 8), generator=gen, dtype=dtype)


def blend_latents(bg, fg, fg_mask, fg_blending_ratio=0.01):
    """latents.py:156-166: bg (1-M) + (bg sqrt(1-r) + fg sqrt(r)) M  (numpy float64 scalars)"""
    dtype = bg.dtype
    out = bg * (1.0 - fg_mask) + (bg * np.sqrt(1.0 - fg_blending_ratio) + fg * np.sqrt(fg_blending_ratio)) * fg_mask
    return out.to(dtype)


def get_input_latents_list(bg_seed, fg_seed_start, fg_blending_ratio, height, width, so_boxes,
                           in_channels=4, init_noise_sigma=1.0, dtype=torch.float32):
    """latents.py:257-295 -> (input_latents_list, latents_bg, fg_seed_list)"""
    bg = get_unscaled_latents(1, in_channels, height, width, bg_seed, dtype)
    outs, seeds = [], []
    for box in so_boxes:
        H, W = height // 8, width // 8
        m = geo.proportion_to_mask(box, H, W)
        seeds.append(fg_seed_start)
        fg = get_unscaled_latents(1, in_channels, height, width, fg_seed_start, dtype)
        outs.append(blend_latents(bg, fg, m, fg_blending_ratio) * init_noise_sigma)
    return outs, bg * init_noise_sigma, seeds


def get_input_latents_lne(idx, bg_seed, fg_seed_start, fg_blending_ratio, height, width, so_boxes,
                          in_channels=4, init_noise_sigma=1.0, dtype=torch.float32):
    """latents.py:298-325"""
    return get_input_latents_list(bg_seed, fg_seed_start, fg_blending_ratio, height, width,
                                  [so_boxes[idx]], in_channels, init_noise_sigma, dtype)[0][0]


def align_with_bboxes(latents_all_list, mask_list, bboxes, horizontal_shift_only=False):
    """latents.py:220-240: mask centroid -> offset to the box centre -> 1/8-quantised zero-filled shift."""
    new_l, new_m, offs = [], [], []
    for lat, m, bb in zip(latents_all_list, mask_list, bboxes):
        xs, ys = geo.binary_mask_to_center(m, normalize=True)
        x0, y0, x1, y1 = bb
        xo, yo = (x0 + x1) / 2 - xs, (y0 + y1) / 2 - ys
        if horizontal_shift_only:
            yo = 0.0
        new_l.append(geo.shift_tensor(lat, xo, yo, offset_normalized=True))
        new_m.append(geo.shift_tensor(m, xo, yo, offset_normalized=True))
        offs.append((xo, yo))
    return new_l, new_m, offs


def compose_latents(latents_all_list, mask_list, latents_bg, num_steps_plus_one, compose_box_to_bg=True):
    """latents.py:168-218 (use_fast_schedule=False): largest mask first; step 0 pasted inside the BOX mask,
    then all steps inside the segmentation mask; returns (composed [S,1,C,h,w], foreground_indices [h,w] long)."""
    dtype = latents_bg.dtype
    composed = torch.zeros((num_steps_plus_one, *latents_bg.shape), dtype=dtype)
    composed[0] = latents_bg
    fg_idx = torch.zeros(latents_bg.shape[-2:], dtype=torch.long)
    sizes = np.array([m.sum().item() for m in mask_list])
    order = np.argsort(-sizes)
    if compose_box_to_bg:
        for i in order:
            bm = geo.binary_mask_to_box_mask(mask_list[i])[None, None, None, ...].to(dtype)
            composed[0] = composed[0] * (1.0 - bm) + latents_all_list[i][0] * bm
    for i in order:
        m = mask_list[i]
        fg_idx = fg_idx * (~m) + (i + 1) * m
        me = m[None, None, None, ...].to(dtype)
        composed = composed * (1.0 - me) + latents_all_list[i] * me
    return composed, fg_idx

"""Seeded initial latents, fg/bg noise blending, alignment to boxes and masked latent composition.

Mirrors reference ``utils/latents.py``: ``get_unscaled_latents`` :138-149, ``get_scaled_latents`` :151-154,
``blend_latents`` :156-166, ``compose_latents`` :168-218, ``align_with_bboxes`` :220-240,
``get_input_latents_list`` :257-295, ``get_input_latents_lne`` :298-325.  Same signatures (``adapter`` = anything with
``.pipe.unet.config.in_channels`` / ``.pipe.unet.dtype`` / ``.pipe.scheduler.init_noise_sigma``).

RNG parity is host-side by construction, exactly like the reference: noise is drawn from the CPU generator seeded
by ``torch.manual_seed`` (:144-147, :263, :284) IN ``unet.dtype`` (an fp16 draw is a different sequence than an fp32
draw; ``generate.py:77-81`` runs fp16) and every object gets the SAME ``fg_seed_start`` (:282-283); only then it moves
to the GPU.  The arithmetic (blend, zero-filled shift, masked paste over all 51 steps) runs in ``tg_blend_latents`` /
``tg_shift`` / ``tg_masked_compose`` on fp32 device tensors; for half-precision latents the blend rounds where the
reference's half-precision tensor ops round, so the result equals the reference's bit for bit (tests/golden/latents_half.npz).
``prepare_mid_image`` (:48-135, pixel-space PIL paste) is outside the hot path.
"""
import numpy as np
import torch

from . import ops, utils

torch_device = "cuda"


def get_unscaled_latents(batch_size, in_channels, height, width, generator, dtype):
    """CPU draw in ``dtype`` (= unet.dtype in the flow), then to the device (reference :138-149)."""
    return torch.randn((batch_size, in_channels, height // 8, width // 8), generator=generator, dtype=dtype).to(torch_device, dtype=dtype)


def get_scaled_latents(batch_size, in_channels, height, width, generator, dtype, scheduler):
    lat = get_unscaled_latents(batch_size, in_channels, height, width, generator, dtype)
    sigma = float(scheduler.init_noise_sigma)
    if sigma == 1.0:
        return lat
    raise NotImplementedError("init_noise_sigma != 1 (non-DDIM schedulers) is not on the hot path")


def blend_latents(latents_bg, latents_fg, fg_mask, fg_blending_ratio=0.01, sigma=1.0):
    """bg (1-M) + (bg sqrt(1-r) + fg sqrt(r)) M, optionally times init_noise_sigma (reference :156-166, :288); result in
    the dtype of ``latents_bg`` with the reference's rounding points for fp16 / bf16 latents."""
    dtype = latents_bg.dtype
    out = ops.blend_latents(latents_bg.to(torch.float32).contiguous(), latents_fg.to(torch.float32).contiguous(),
                            fg_mask.to(device=latents_bg.device, dtype=torch.float32).contiguous(), fg_blending_ratio, sigma,
                            storage_dtype=dtype if dtype in (torch.float16, torch.bfloat16) else None)
    return out.to(dtype)


def get_input_latents_list(model_dict, bg_seed, fg_seed_start, fg_blending_ratio, height, width, adapter,
                           so_prompt_phrase_box_list=None, so_boxes=None, verbose=False):
    """-> (input_latents_list, latents_bg, fg_seed_list), all scaled by init_noise_sigma (reference :257-295)."""
    unet, scheduler = adapter.pipe.unet, adapter.pipe.scheduler
    dtype = unet.dtype                                # the reference draws AND blends in unet.dtype (:261-288)
    sigma = float(scheduler.init_noise_sigma)
    latents_bg = get_unscaled_latents(1, unet.config.in_channels, height, width, torch.manual_seed(bg_seed), dtype)
    if so_boxes is None:
        so_boxes = [item[-1] for item in so_prompt_phrase_box_list]
    input_latents_list, fg_seed_list = [], []
    H, W = height // 8, width // 8
    for obj_box in so_boxes:
        fg_mask = utils.proportion_to_mask(obj_box, H, W, device=latents_bg.device)
        fg_seed = fg_seed_start                       # the reference gives every object the same fg seed (:282)
        fg_seed_list.append(fg_seed)
        latents_fg = get_unscaled_latents(1, unet.config.in_channels, height, width, torch.manual_seed(fg_seed), dtype)
        input_latents_list.append(blend_latents(latents_bg, latents_fg, fg_mask, fg_blending_ratio, sigma))
    if sigma != 1.0:
        raise NotImplementedError("init_noise_sigma != 1 is not on the hot path")
    return input_latents_list, latents_bg, fg_seed_list


def get_input_latents_lne(idx, adapter, model_dict, bg_seed, fg_seed_start, fg_blending_ratio, height, width,
                          so_prompt_phrase_box_list=None, so_boxes=None, verbose=False):
    if so_boxes is None:
        so_boxes = [item[-1] for item in so_prompt_phrase_box_list]
    return get_input_latents_list(model_dict, bg_seed, fg_seed_start, fg_blending_ratio, height, width, adapter,
                                  so_boxes=[so_boxes[idx]])[0][0]


def align_with_bboxes(latents_all_list, mask_tensor_list, bboxes, horizontal_shift_only=False):
    """Mask centroid -> offset to the box centre -> 1/8-quantised zero-filled shift of the [51,1,C,h,w] latents
    (GPU) and of the [h,w] mask (host) — reference :220-240."""
    new_latents, new_masks, offsets = [], [], []
    for latents_all, mask, bbox in zip(latents_all_list, mask_tensor_list, bboxes):
        xs, ys = utils.binary_mask_to_center(mask, normalize=True)
        x0, y0, x1, y1 = bbox
        x_off, y_off = (x0 + x1) / 2 - xs, (y0 + y1) / 2 - ys
        if horizontal_shift_only:
            y_off = 0.0
        new_latents.append(utils.shift_tensor(latents_all, x_off, y_off, offset_normalized=True))
        new_masks.append(utils.shift_tensor(mask, x_off, y_off, offset_normalized=True))
        offsets.append((x_off, y_off))
    return new_latents, new_masks, offsets


@torch.no_grad()
def compose_latents(adapter, model_dict, latents_all_list, mask_tensor_list, num_inference_steps, overall_batch_size,
                    height, width, latents_bg=None, bg_seed=None, compose_box_to_bg=True, use_fast_schedule=False,
                    fast_after_steps=None):
    """Largest mask first: step-0 latents pasted inside each object's BOX mask, then all steps inside its
    segmentation mask; returns (composed [S,1,C,h,w] fp32 on the GPU, foreground_indices [h,w] long) — reference :168-218."""
    unet, scheduler = adapter.pipe.unet, adapter.pipe.scheduler
    if latents_bg is None:
        # drawn in unet.dtype like the reference (:170-173); the masked paste below only copies values (masks are 0 / 1), so
        # the fp32 result holds exactly the reference's half-precision values
        latents_bg = get_scaled_latents(overall_batch_size, unet.config.in_channels, height, width, torch.manual_seed(bg_seed),
                                        unet.dtype, scheduler)
    dev = latents_bg.device
    n_rows = (fast_after_steps + 1) if use_fast_schedule else (num_inference_steps + 1)
    composed = torch.zeros((n_rows, *latents_bg.shape), dtype=torch.float32, device=dev)
    composed[0].copy_(latents_bg)
    fg_idx = torch.zeros(latents_bg.shape[-2:], dtype=torch.long)
    masks_host = [m.detach().cpu() for m in mask_tensor_list]
    order = np.argsort(-np.array([m.sum().item() for m in masks_host]))
    if compose_box_to_bg:
        for i in order:
            box_mask = utils.binary_mask_to_box_mask(masks_host[i], to_device=False).to(dev)
            ops.masked_compose_(composed[0], latents_all_list[i][0].to(torch.float32).contiguous(), box_mask)
    for i in order:
        m = masks_host[i].to(torch.bool)
        fg_idx = fg_idx * (~m) + (i + 1) * m                      # 64x64 label map: host integer work
        src = latents_all_list[i][:n_rows].to(device=dev, dtype=torch.float32).contiguous()
        ops.masked_compose_(composed, src, m.to(device=dev, dtype=torch.float32))
    return composed, fg_idx.to(dev)

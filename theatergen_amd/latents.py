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
``prepare_mid_image`` (:48-135) and ``compose_latents_with_alignment`` (:242-255) close the stage-1 -> stage-2 hand-off of
``theatergen.py:415-423``: the paste is host-side integer work on 512 x 512 uint8 pixels (PIL resize + numpy), run once per turn, so it
stays on the host — bit-exact against the imported reference (tests/golden/mid_image.npz); the latent side runs on the kernels above.
"""
import os

import numpy as np
import torch

from . import ops, utils

torch_device = "cuda"
MID_IMAGE_DIR = "visualization"      # the reference writes its two PNGs here (utils/latents.py:133-134); None switches the side effect off


def _mask_extent(mask):
    """(first row, first column, last row, last column) of the set pixels of a 2-D mask, or None when it is empty
    (reference ``find_bounding_box`` :32-41; the last row / column are INCLUSIVE, the crop below is not)."""
    m = np.asarray(mask.detach().cpu() if torch.is_tensor(mask) else mask) != 0
    rows, cols = np.flatnonzero(m.any(axis=1)), np.flatnonzero(m.any(axis=0))
    if rows.size == 0:
        return None
    return int(rows[0]), int(cols[0]), int(rows[-1]), int(cols[-1])


def prepare_mid_image(basever, repeat_ind, mask_tensor_list_512, single_obj_img_list, bboxes):
    """Paste every character's segmented crop, rescaled to fit its layout box, onto one black canvas and build the inpainting
    mask of stage 2 -> (mask ``PIL 'L'``: 0 inside the pasted characters, 255 elsewhere; canvas ``PIL 'RGB'``).  Reference
    ``utils/latents.py:48-135``; caller ``theatergen.py:415-423`` -> ``models/pipelines.py:605-631``.

    The reference's arithmetic is uint8 throughout and this restatement keeps its consequences: a character's crop is the mask's
    bounding box WITHOUT its last row / column (:68-69, :73-74); the fit factor is the larger of the two aspect quotients (:82); the crop
    and its mask go through ``PIL.Image.resize`` at the default filter and the mask is re-binarised (:86-92); an earlier character wins
    over a later one where they overlap (:110-125); where two masks overlap the mask sum wraps (255 + 255 = 254 in uint8, :128-129), which makes
    both cover factors 0 there — the pixel ends black and the returned mask holds 1, not 0."""
    from PIL import Image
    first = mask_tensor_list_512[0]
    m, n = int(first.shape[0]), int(first.shape[1])
    pasted = np.zeros((m, n), dtype=np.uint8)                 # 255 where a character sits (sum of the pasted masks, uint8 wrap-around)
    canvas = np.zeros((n, m, 3), dtype=np.uint8)              # the reference's Image.new('RGB', (m, n)) is n rows x m columns
    for image, mask, box in zip(single_obj_img_list, mask_tensor_list_512, bboxes):
        x0, y0, x1, y1 = box[0], box[1], box[2], box[3]
        mh, mw = int(mask.shape[0]), int(mask.shape[1])
        box_w_px, box_h_px = abs(x1 - x0) * mw, abs(y1 - y0) * mh
        left, top = int(x0 * mw), int(y0 * mh)
        r0, c0, r1, c1 = _mask_extent(mask)
        crop_w, crop_h = abs(c1 - c0), abs(r1 - r0)
        mask_np = np.asarray(mask.detach().cpu() if torch.is_tensor(mask) else mask)
        crop_img = Image.fromarray(np.asarray(image)[r0:r1, c0:c1, :])
        crop_msk = Image.fromarray(np.where(mask_np[r0:r1, c0:c1], 255, 0).astype(np.uint8), mode="L")
        fit = max(crop_w / box_w_px, crop_h / box_h_px)
        new_w, new_h = int(crop_w / fit), int(crop_h / fit)
        small_msk = np.array(crop_msk.resize((new_w, new_h)))
        small_msk[small_msk > 0] = 255
        small_img = np.array(crop_img.resize((new_w, new_h))) * (small_msk // 255)[..., None]
        # destination window, clipped by the canvas (:118-121)
        dest = canvas[top:top + new_h, left:left + new_w]
        dh, dw = dest.shape[0], dest.shape[1]
        small_img, small_msk = small_img[:dh, :dw], small_msk[:dh, :dw]
        keep_new = ((~pasted) / 255).astype(np.uint8)          # 1 where nothing was pasted before (255 / 255), 0 elsewhere — incl. wrapped 254s
        keep_old = (pasted / 255).astype(np.uint8)
        before = canvas.copy()
        canvas[top:top + new_h, left:left + new_w] = small_img
        canvas = canvas * keep_new[..., None] + before * keep_old[..., None]
        pasted[top:top + new_h, left:left + new_w] += small_msk
    new_mask = Image.fromarray(~pasted, mode="L")
    white_image = Image.fromarray(canvas)
    if MID_IMAGE_DIR is not None:
        os.makedirs(MID_IMAGE_DIR, exist_ok=True)
        white_image.save(os.path.join(MID_IMAGE_DIR, f"{repeat_ind}vis_image.png"))
        new_mask.save(os.path.join(MID_IMAGE_DIR, f"{repeat_ind}vis_mask.png"))
    return new_mask, white_image


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


@torch.no_grad()
def compose_latents_with_alignment(basever, adapter, repeat_ind, mask_tensor_list_512, single_obj_img_list, model_dict, latents_all_list,
                                   mask_tensor_list, num_inference_steps, overall_batch_size, height, width,
                                   align_with_overall_bboxes=True, overall_bboxes=None, horizontal_shift_only=False, **kwargs):
    """The stage-1 -> stage-2 hand-off (reference ``utils/latents.py:242-255``, called from ``theatergen.py:415-423``): shift every
    character's 51-step latents and 64 x 64 mask onto its layout box (``align_with_bboxes``, ``tg_shift``), paste the pixel-space crops
    (``prepare_mid_image``), then compose the latents (``compose_latents``, ``tg_masked_compose``) ->
    ``(composed_latents, foreground_indices, inp_mask, inp_img)``.  As in the reference the pasted image only exists on the aligned,
    non-empty path; the other path raises (there: ``UnboundLocalError`` on the return statement, here: a ``RuntimeError`` that says why,
    which the caller's policy ``generate.py:250-259`` treats the same way — skip the turn)."""
    pasted = None
    if align_with_overall_bboxes and len(latents_all_list):
        flat_boxes = utils.expand_overall_bboxes(overall_bboxes)
        latents_all_list, mask_tensor_list, _ = align_with_bboxes(latents_all_list, mask_tensor_list, bboxes=flat_boxes,
                                                                  horizontal_shift_only=horizontal_shift_only)
        pasted = prepare_mid_image(basever, repeat_ind, mask_tensor_list_512, single_obj_img_list, bboxes=flat_boxes)
    composed, fg_idx = compose_latents(adapter, model_dict, latents_all_list, mask_tensor_list, num_inference_steps, overall_batch_size,
                                       height, width, **kwargs)
    if pasted is None:
        raise RuntimeError("compose_latents_with_alignment: no pasted mid image (align_with_overall_bboxes is off or there is no character); "
                           "the reference has no return value on this path either (utils/latents.py:250-255)")
    return composed, fg_idx, pasted[0], pasted[1]

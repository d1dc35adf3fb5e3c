"""Synthetic CMIGBench 4-turn story workload (no datasets / checkpoints exist offline).

Per dialogue d, turn tau in 1..4, character c in {0, 1}: one stage-1 character generation
(reference ``theatergen.py:204-273`` -> ``generate_single_object_with_box`` :43-201 ->
``pipelines.generate_semantic_guidance``) with
  * initial latents from the reference recipe (``utils/latents.py:257-295``): bg_seed = dialogue seed offset,
    fg_seed_start = bg_seed + 123456789 (``generate.py:157, 236-243``), blended inside the character box,
  * boxes [x, y, w, h] / 512 = [40,150,190,300] and [280,150,190,300],
  * text embeddings ``randn(1, 77, D) * 0.5`` seeded by (d, tau, c); negative-prompt embeddings seeded once (shared),
  * image tokens ``randn(1, T, D) * 0.5`` seeded by the CHARACTER id (shared across turns: the broadcast payload);
    uncond image tokens seeded once,
  * IP scale 0.4, guidance 7.5, 50 DDIM steps.
"""
from dataclasses import dataclass

import torch

BOXES_XYWH = ([40, 150, 190, 300], [280, 150, 190, 300], [150, 40, 120, 120], [330, 40, 120, 120])
FG_SEED_OFFSET = 123456789


def box_xyxy(i, size=512):
    x, y, w, h = BOXES_XYWH[i]
    return [x / size, y / size, (x + w) / size, (y + h) / size]


@dataclass(frozen=True)
class CharacterJob:
    dialogue: int
    turn: int
    char: int

    @property
    def char_id(self):
        return self.dialogue * 16 + self.char          # characters persist across the turns of one dialogue

    @property
    def text_seed(self):
        return 1_000_003 * self.dialogue + 101 * self.turn + self.char + 17

    @property
    def bg_seed(self):
        return self.dialogue * 100 + self.turn          # seed_offset per (dialogue, turn)


def story_jobs(dialogue, turns=4, chars=2):
    return [CharacterJob(dialogue, t, c) for t in range(1, turns + 1) for c in range(chars)]


def _randn(shape, seed, scale=0.5):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def shared_conditioning(ctx, num_tokens, dtype, device, text_len=77):
    """Tensors rank 0 generates and broadcasts: negative text embeds and the uncond image tokens."""
    return {"neg_text": _randn((1, text_len, ctx), 7).to(device, dtype),
            "uncond_image": _randn((1, num_tokens, ctx), 11).to(device, dtype)}


def character_image_tokens(char_ids, ctx, num_tokens, dtype, device):
    """[n_chars, T, ctx] image tokens, one row per character id (what IPAdapter.get_image_embeds would return)."""
    return torch.stack([_randn((num_tokens, ctx), 5000 + cid) for cid in char_ids]).to(device, dtype)


def job_conditioning(jobs, shared, image_tokens, char_index, ctx, dtype, device, text_len=77):
    """encoder_hidden_states [2n, 77+T, ctx]: rows [0, n) = negative (text ; uncond image), rows [n, 2n) = positive."""
    n = len(jobs)
    text = torch.stack([_randn((text_len, ctx), j.text_seed) for j in jobs]).to(device, dtype)
    img = torch.stack([image_tokens[char_index[j.char_id]] for j in jobs])
    pos = torch.cat([text, img], dim=1)
    neg = torch.cat([shared["neg_text"].expand(n, -1, -1), shared["uncond_image"].expand(n, -1, -1)], dim=1)
    return torch.cat([neg, pos], dim=0).contiguous()


def job_latents(jobs, adapter, height=512, width=512, fg_blending_ratio=0.01):
    """[n, 4, h/8, w/8] fp32 initial latents via the reference recipe (CPU RNG, blend on the GPU)."""
    from . import latents as L
    outs = []
    for j in jobs:
        lst, _, _ = L.get_input_latents_list(None, j.bg_seed, j.bg_seed + FG_SEED_OFFSET, fg_blending_ratio, height, width,
                                             adapter, so_boxes=[box_xyxy(j.char)])
        outs.append(lst[0])
    return torch.cat(outs, dim=0).to(torch.float32)

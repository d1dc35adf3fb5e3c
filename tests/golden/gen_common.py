"""Seeded input / weight builders shared by ``make_golden.py`` (which runs the imported REFERENCE in the
build container) and by the tests (which re-create the same inputs for the oracle / the HIP path).
Only torch CPU generators are used, so the tensors are reproducible wherever the same torch runs.
"""
import math

import torch

# (name, C, heads, ctx, N, T)  -- head dim = C // heads: 40 / 80 / 160 (SD-1.5) and 64 (SD-2.1 / SDXL)
ATTN_CASES = [
    ("sd15_c320", 320, 8, 768, 64, 4),
    ("sd15_c640", 640, 8, 768, 36, 4),
    ("sd15_c1280", 1280, 8, 768, 16, 16),
    ("sd21_c640", 640, 10, 1024, 36, 4),
    ("sdxl_c1280", 1280, 20, 2048, 16, 16),
]
IP_SCALES = [0.0, 0.1, 0.4, 1.0]

RESAMPLER_CASES = {
    # reference ip_adapter/test_resampler.py:18-30 shape family (pos-emb + mean-pooled latents), small dims
    "small_posemb": dict(dim=64, depth=2, dim_head=16, heads=4, num_queries=8, embedding_dim=48, output_dim=80,
                         ff_mult=4, max_seq_len=33, apply_pos_emb=True, num_latents_mean_pooled=4, seq=33),
    # IPAdapterPlus.init_proj, reference ip_adapter/ip_adapter.py:292-303
    "sd15_plus": dict(dim=768, depth=4, dim_head=64, heads=12, num_queries=16, embedding_dim=1280, output_dim=768,
                      ff_mult=4, seq=257),
    # IPAdapterPlusXL.init_proj, reference ip_adapter/ip_adapter.py:334-345
    "sdxl_plus": dict(dim=1280, depth=4, dim_head=64, heads=20, num_queries=16, embedding_dim=1280, output_dim=2048,
                      ff_mult=4, seq=257),
}


def _u(shape, fan_in, g):
    return (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)


def attn_weights(C, ctx, seed, with_ip=True):
    """Parameter dict with the reference's names (Attention :113-128, IPAttnProcessor :418-419)."""
    g = torch.Generator().manual_seed(seed)
    w = {
        "to_q.weight": _u((C, C), C, g),
        "to_k.weight": _u((C, ctx), ctx, g),
        "to_v.weight": _u((C, ctx), ctx, g),
        "to_out.0.weight": _u((C, C), C, g),
        "to_out.0.bias": _u((C,), C, g),
    }
    if with_ip:
        w["to_k_ip.weight"] = _u((C, ctx), ctx, g)
        w["to_v_ip.weight"] = _u((C, ctx), ctx, g)
    return w


def attn_inputs(C, ctx, N, T, seed, batch=2, text_len=77):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((batch, N, C), generator=g)
    enc = torch.randn((batch, text_len + T, ctx), generator=g) * 0.5
    return x, enc


def case_scales(ci):
    """all four IP scales on the first case, scale 0.4 elsewhere (keeps the fixtures small)"""
    return IP_SCALES if ci == 0 else [0.4]


def imageproj_params(seed=600):
    """ImageProjModel (reference ip_adapter/ip_adapter.py:30-47): Linear(1024 -> 4*768) + LayerNorm(768)."""
    g = torch.Generator().manual_seed(seed)
    sd = {"proj.weight": _u((4 * 768, 1024), 1024, g), "proj.bias": _u((4 * 768,), 1024, g),
          "norm.weight": 1 + 0.1 * torch.randn(768, generator=g), "norm.bias": 0.1 * torch.randn(768, generator=g)}
    e = torch.randn(2, 1024, generator=g)
    return sd, e


def mlpproj_params(seed=610, clip_dim=1280, ctx=768):
    """MLPProjModel (reference ip_adapter/ip_adapter.py:50-64): Linear -> GELU -> Linear -> LayerNorm (IP-Adapter-Full)."""
    g = torch.Generator().manual_seed(seed)
    sd = {"proj.0.weight": _u((clip_dim, clip_dim), clip_dim, g), "proj.0.bias": _u((clip_dim,), clip_dim, g),
          "proj.2.weight": _u((ctx, clip_dim), clip_dim, g), "proj.2.bias": _u((ctx,), clip_dim, g),
          "proj.3.weight": 1 + 0.1 * torch.randn(ctx, generator=g), "proj.3.bias": 0.1 * torch.randn(ctx, generator=g)}
    e = torch.randn(2, 257, clip_dim, generator=g)
    return sd, e


GUIDANCE_KEYS = [("mid", 0, 0, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 1, 2, 0)]
GUIDANCE_HW = {GUIDANCE_KEYS[0]: 64, GUIDANCE_KEYS[1]: 256, GUIDANCE_KEYS[2]: 256, GUIDANCE_KEYS[3]: 256}
GUIDANCE_BOXES = {
    1: [[40 / 512, 150 / 512, 230 / 512, 450 / 512]],
    2: [[40 / 512, 150 / 512, 230 / 512, 450 / 512], [280 / 512, 150 / 512, 470 / 512, 450 / 512]],
    4: [[40 / 512, 150 / 512, 230 / 512, 450 / 512], [280 / 512, 150 / 512, 470 / 512, 450 / 512],
        [150 / 512, 40 / 512, 270 / 512, 160 / 512], [[330 / 512, 40 / 512, 450 / 512, 160 / 512], [0.0, 0.0, 0.12, 0.12]]],
}
GUIDANCE_POSITIONS = {1: [[2, 3]], 2: [[2, 3], [7]], 4: [[2, 3], [7], [10, 11, 12], [15]]}


def guidance_attn_maps(nbox):
    """Synthetic cond-half attention maps [1, heads=8, HW, 77] whose rows sum to one (softmax-like)."""
    g = torch.Generator().manual_seed(800 + nbox)
    maps = {}
    for k in GUIDANCE_KEYS:
        a = torch.rand(1, 8, GUIDANCE_HW[k], 77, generator=g)
        maps[k] = a / a.sum(-1, keepdim=True)
    return maps, g


def guidance_ref_maps(g):
    return [{3: {k: torch.rand(1, 8, GUIDANCE_HW[k], 1, generator=g) for k in GUIDANCE_KEYS}} for _ in range(2)]


def resampler_input(case, seed, batch=2):
    g = torch.Generator().manual_seed(seed)
    return torch.randn((batch, case["seq"], case["embedding_dim"]), generator=g)


class FakeTokenizer:
    """Whitespace tokenizer with <bos>/<eos>, standing in for the CLIP tokenizer (vocab unavailable
    offline) in ``get_phrase_indices`` tests; implements just what reference utils/guidance.py:10-30 uses."""
    eos_token = "<eos>"
    bos_token = "<bos>"

    def __init__(self):
        self.vocab = {}
        self.inv = {}

    def _id(self, tok):
        if tok not in self.vocab:
            self.vocab[tok] = len(self.vocab)
            self.inv[self.vocab[tok]] = tok
        return self.vocab[tok]

    def __call__(self, prompts, padding="do_not_pad", max_length=77, return_tensors="np"):
        import numpy as np
        out = []
        for p in prompts:
            toks = [self.bos_token] + p.replace("|", " | ").replace(",", " , ").split() + [self.eos_token]
            out.append(np.array([self._id(t) for t in toks]))
        return {"input_ids": out}

    def _convert_id_to_token(self, i):
        return self.inv[i]


# ---- round 4: the pre-projection branches of AttnProcessor (reference ip_adapter/attention_processor.py:316-347) ----
# (name, ctor kwargs of Attention, cross?, mask kind, 4-D input?)
BRANCH_C, BRANCH_HEADS, BRANCH_CTX, BRANCH_N, BRANCH_L = 128, 4, 96, 64, 24
BRANCH_CASES = [
    ("gn_self4d", dict(norm_num_groups=8, residual_connection=True, rescale_output_factor=2.0), False, None, True),
    ("gn_self_bias", dict(norm_num_groups=16, bias=True), False, None, False),
    ("lnx_cross", dict(cross_attention_norm="layer_norm"), True, None, False),
    ("gnx_cross", dict(cross_attention_norm="group_norm", cross_attention_norm_num_groups=8), True, None, False),
    ("mask_self_b1l", dict(), False, "b1l", False),
    ("mask_self_bql", dict(), False, "bql", False),
    ("mask_cross_b1l", dict(bias=True), True, "b1l", False),
    ("mask_cross_bhql", dict(), True, "bhql", False),
]


def branch_params(name, kw, cross, seed):
    """state dict (reference parameter names) + inputs of one branch case"""
    g = torch.Generator().manual_seed(seed)
    C, ctx = BRANCH_C, (BRANCH_CTX if cross else BRANCH_C)
    sd = {"to_q.weight": _u((C, C), C, g), "to_k.weight": _u((C, ctx), ctx, g), "to_v.weight": _u((C, ctx), ctx, g),
          "to_out.0.weight": _u((C, C), C, g), "to_out.0.bias": _u((C,), C, g)}
    if kw.get("bias"):
        for n in ("to_q", "to_k", "to_v"):
            sd[n + ".bias"] = 0.3 * torch.randn(C, generator=g)
    if kw.get("norm_num_groups"):
        sd["group_norm.weight"] = 1 + 0.2 * torch.randn(C, generator=g)
        sd["group_norm.bias"] = 0.2 * torch.randn(C, generator=g)
    if kw.get("cross_attention_norm"):
        sd["norm_cross.weight"] = 1 + 0.2 * torch.randn(ctx, generator=g)
        sd["norm_cross.bias"] = 0.2 * torch.randn(ctx, generator=g)
    x = torch.randn(2, BRANCH_N, C, generator=g) * 1.5 + 0.5
    enc = (torch.randn(2, BRANCH_L, ctx, generator=g) * 0.7 + 0.2) if cross else None
    return sd, x, enc


def branch_mask(kind, cross, seed):
    """additive bias masks in the form UNet2DConditionModel builds them ((1 - m) * -10000, models/unet_2d_condition.py:785-796) and
    the denser forms baddbmm accepts"""
    if kind is None:
        return None
    g = torch.Generator().manual_seed(seed)
    L = BRANCH_L if cross else BRANCH_N
    if kind == "b1l":
        keep = (torch.rand(2, 1, L, generator=g) > 0.3).float()
        keep[..., 0] = 1
        return (1 - keep) * -10000.0
    if kind == "bql":
        return torch.randn(2, BRANCH_N, L, generator=g) * 2
    if kind == "bhql":
        return torch.randn(2 * BRANCH_HEADS, BRANCH_N, L, generator=g) * 2
    raise ValueError(kind)


# ---- round 5: SD-1.5 FIRST-LEVEL geometry (320 channels = 8 heads x 40, 77 + T tokens) composed the way models/attention.py:186-236 and
# models/transformer_2d.py:285-327 compose the reference's parts, so that the fixtures route through the row-chain kernels (tg_rc_xattn,
# tg_rc_linear, tg_rc_ff, tg_rc_front) when the tests lower rowchain.MIN_ROWS / MIN_ROWS_CHAIN
BLOCK_C, BLOCK_HEADS, BLOCK_CTX, BLOCK_B, BLOCK_H, BLOCK_W = 320, 8, 768, 2, 8, 16        # 128 tokens per item = one 128-row workgroup each
BLOCK_T = [4, 16]


def block_params(T, seed=900):
    """state dict of a Transformer2DModel(in_channels 320, 1 layer, conv 1x1 projections: SD-1.5) with the reference's parameter names
    (+ ``attn2.processor.to_{k,v}_ip``), input x [2, 320, 8, 16] and encoder states [2, 77 + T, 768]"""
    g = torch.Generator().manual_seed(seed + T)
    C, ctx = BLOCK_C, BLOCK_CTX
    sd = {"norm.weight": 1 + 0.2 * torch.randn(C, generator=g), "norm.bias": 0.1 * torch.randn(C, generator=g),
          "proj_in.weight": _u((C, C, 1, 1), C, g), "proj_in.bias": _u((C,), C, g),
          "proj_out.weight": _u((C, C, 1, 1), C, g), "proj_out.bias": _u((C,), C, g)}
    b = "transformer_blocks.0."
    for n in ("norm1", "norm2", "norm3"):
        sd[b + n + ".weight"] = 1 + 0.2 * torch.randn(C, generator=g)
        sd[b + n + ".bias"] = 0.1 * torch.randn(C, generator=g)
    for a, kdim in (("attn1", C), ("attn2", ctx)):
        sd[b + a + ".to_q.weight"] = _u((C, C), C, g)
        sd[b + a + ".to_k.weight"] = _u((C, kdim), kdim, g)
        sd[b + a + ".to_v.weight"] = _u((C, kdim), kdim, g)
        sd[b + a + ".to_out.0.weight"] = _u((C, C), C, g)
        sd[b + a + ".to_out.0.bias"] = _u((C,), C, g)
    sd[b + "attn2.processor.to_k_ip.weight"] = _u((C, ctx), ctx, g)
    sd[b + "attn2.processor.to_v_ip.weight"] = _u((C, ctx), ctx, g)
    sd[b + "ff.net.0.proj.weight"] = _u((8 * C, C), C, g)
    sd[b + "ff.net.0.proj.bias"] = _u((8 * C,), C, g)
    sd[b + "ff.net.2.weight"] = _u((C, 4 * C), 4 * C, g)
    sd[b + "ff.net.2.bias"] = _u((C,), 4 * C, g)
    x = torch.randn(BLOCK_B, C, BLOCK_H, BLOCK_W, generator=g) * 1.2 + 0.3
    enc = torch.randn(BLOCK_B, 77 + T, ctx, generator=g) * 0.5
    return sd, x, enc


# ---- round 5: inner-level cross-attention sub-block geometry (32 x 32 level: 640 = 8 x 80; 16 x 16 level: 1280 = 8 x 160) with whole 128-token tiles per
# batch item, so that the fixtures route through the fused norm2 + to_q + attention launch (tg_xq_attn).  (name, C, heads, ctx, N, T, scale, ip?)
XQ_CASES = [
    ("c640_ip4", 640, 8, 768, 128, 4, 0.4, True),
    ("c640_plain", 640, 8, 768, 128, 0, 0.0, False),
    ("c1280_ip16", 1280, 8, 768, 128, 16, 1.0, True),
]


def xq_params(ci):
    name, C, heads, ctx, N, T, scale, ip = XQ_CASES[ci]
    g = torch.Generator().manual_seed(1100 + ci)
    w = attn_weights(C, ctx, seed=1200 + ci, with_ip=ip)
    norm = {"weight": 1 + 0.2 * torch.randn(C, generator=g), "bias": 0.1 * torch.randn(C, generator=g)}
    x = torch.randn(2, N, C, generator=g) * 1.2 + 0.3
    enc = torch.randn(2, 77 + T, ctx, generator=g) * 0.5
    return w, norm, x, enc


def mid_image_case(case):
    """Synthetic inputs of ``prepare_mid_image`` (reference utils/latents.py:48-135): per character a 512 x 512 segmentation mask (an ellipse with a notch,
    so the bounding box is not the mask), a deterministic RGB image with gradients and a checker pattern (exercises the resampling filter, compresses
    well) and a normalised layout box.  case 0: two separate characters; 1: overlapping boxes (the uint8 wrap of the summed masks); 2: a box that leaves
    the canvas (destination clipping) + a third small character."""
    import numpy as np
    yy, xx = np.mgrid[0:512, 0:512]
    specs = {
        0: [((150, 200), (90, 140), [40 / 512, 150 / 512, 230 / 512, 450 / 512]), ((360, 260), (70, 170), [280 / 512, 150 / 512, 470 / 512, 450 / 512])],
        1: [((200, 250), (120, 160), [0.15, 0.2, 0.6, 0.9]), ((300, 260), (110, 150), [0.4, 0.25, 0.85, 0.95])],
        2: [((256, 256), (200, 120), [0.55, 0.5, 1.1, 0.95]), ((120, 140), (60, 90), [0.05, 0.05, 0.3, 0.45]), ((400, 100), (40, 40), [0.7, 0.05, 0.95, 0.3])],
    }[case]
    masks, images, boxes = [], [], []
    for k, ((cx, cy), (rx, ry), box) in enumerate(specs):
        m = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0
        m &= ~((xx > cx + rx // 3) & (yy < cy - ry // 2))           # notch
        masks.append(torch.from_numpy(m))
        img = np.stack([(xx * 3 + yy * 5 + 40 * c + 17 * ((xx // 16 + yy // 16 + k) % 2) + 29 * k) % 256 for c in range(3)], axis=-1).astype(np.uint8)
        images.append(img)
        boxes.append(box)
    return masks, images, boxes

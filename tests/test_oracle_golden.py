"""CPU: the oracle (oracle/*.py) must reproduce the golden vectors captured from the IMPORTED reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import attention as oattn
from oracle import box_geometry as geo
from oracle import guidance_loss as og
from oracle import latent_ops as ol
from oracle import resampler as ores
from oracle import unet as ounet
from tests.golden import gen_common as gc
from theatergen_amd import weights as W

TOL = dict(rtol=1e-5, atol=2e-6)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _branch_call(ci):
    """(oracle kwargs, inputs) of gc.BRANCH_CASES[ci] — shared with the GPU test"""
    name, kw, cross, mkind, four_d = gc.BRANCH_CASES[ci]
    sd, x, enc = gc.branch_params(name, kw, cross, seed=700 + ci)
    mask = gc.branch_mask(mkind, cross, seed=800 + ci)
    xin = x
    if four_d:
        h = int(gc.BRANCH_N ** 0.5)
        xin = x.transpose(1, 2).reshape(2, gc.BRANCH_C, h, h).contiguous()
    okw = {k: v for k, v in kw.items() if k != "bias"}
    return name, sd, xin, enc, mask, okw


@pytest.mark.parametrize("ci", range(len(gc.BRANCH_CASES)))
def test_attention_processor_branches(golden_dir, ci):
    """group_norm / q-k-v bias / norm_cross / attention_mask branches of AttnProcessor (reference :316-347) vs the imported reference"""
    gold = _load(golden_dir, "attn_branches")
    name, sd, xin, enc, mask, okw = _branch_call(ci)
    got = oattn.attn_processor(sd, gc.BRANCH_HEADS, xin, enc, attention_mask=mask, **okw)
    np.testing.assert_allclose(got.numpy(), gold[f"{name}.out"], rtol=2e-5, atol=4e-6)


@pytest.mark.parametrize("ci", range(len(gc.ATTN_CASES)))
def test_attention_processors(golden_dir, ci):
    gold = _load(golden_dir, "attn")
    name, C, heads, ctx, N, T = gc.ATTN_CASES[ci]
    w = gc.attn_weights(C, ctx, seed=100 + ci)
    ws = gc.attn_weights(C, C, seed=300 + ci, with_ip=False)
    x, enc = gc.attn_inputs(C, ctx, N, T, seed=200 + ci)
    np.testing.assert_allclose(oattn.attn_processor(ws, heads, x).numpy(), gold[f"{name}.self"], **TOL)
    for s in gc.case_scales(ci):
        got = oattn.ip_attn_processor(w, heads, x, enc, s, T).numpy()
        np.testing.assert_allclose(got, gold[f"{name}.ip.scale{s}"], **TOL)
    np.testing.assert_allclose(oattn.cn_attn_processor(w, heads, x, enc, T).numpy(), gold[f"{name}.cn"], **TOL)
    _, p = oattn.ip_attn_processor(w, heads, x, enc, 0.4, T, return_probs=True, return_token_ca_only=5,
                                   return_cond_ca_only=True)
    np.testing.assert_allclose(p.numpy(), gold[f"{name}.cap.int5"], **TOL)
    _, p = oattn.ip_attn_processor(w, heads, x, enc, 0.4, T, return_probs=True,
                                   return_token_ca_only=torch.tensor([1, 3, 7]), return_cond_ca_only=True)
    np.testing.assert_allclose(p.numpy(), gold[f"{name}.cap.idx137"], **TOL)
    if ci == 1:
        h = int(N ** 0.5)
        x4 = x.transpose(1, 2).reshape(2, C, h, h).contiguous()
        got = oattn.ip_attn_processor(w, heads, x4, enc, 0.4, T, residual_connection=True, rescale_output_factor=2.0)
        np.testing.assert_allclose(got.numpy(), gold[f"{name}.ip.4d"], **TOL)


@pytest.mark.parametrize("ci,name", list(enumerate(gc.RESAMPLER_CASES)))
def test_resampler(golden_dir, ci, name):
    gold = _load(golden_dir, "resampler")
    case = gc.RESAMPLER_CASES[name]
    kw = {k: v for k, v in case.items() if k != "seq"}
    sd = W.random_resampler_state_dict(seed=400 + ci, **kw)
    x = gc.resampler_input(case, seed=500 + ci)
    args = (kw["depth"], kw["heads"], kw["dim_head"], kw.get("num_latents_mean_pooled", 0))
    np.testing.assert_allclose(ores.resampler_forward(sd, x, *args).numpy(), gold[f"{name}.out"], rtol=2e-5, atol=5e-6)
    np.testing.assert_allclose(ores.resampler_forward(sd, torch.zeros_like(x), *args).numpy(), gold[f"{name}.zero"],
                               rtol=2e-5, atol=5e-6)


def test_image_proj(golden_dir):
    """imageproj.npz: outputs of the REAL ImageProjModel / MLPProjModel classes (reference ip_adapter/ip_adapter.py:30-64,
    class source executed by the generator); resampler.npz keeps the earlier 3-line restatement — they must agree"""
    gold, old = _load(golden_dir, "imageproj"), _load(golden_dir, "resampler")
    assert np.array_equal(gold["imageproj.out"], old["imageproj.out"]) and np.array_equal(gold["imageproj.zero"], old["imageproj.zero"])
    sd, e = gc.imageproj_params()
    np.testing.assert_allclose(ores.image_proj_model(sd, e, 4, 768).numpy(), gold["imageproj.out"], **TOL)
    np.testing.assert_allclose(ores.image_proj_model(sd, torch.zeros_like(e), 4, 768).numpy(), gold["imageproj.zero"], **TOL)
    sd2, e2 = gc.mlpproj_params()
    np.testing.assert_allclose(ores.mlp_proj_model(sd2, e2).numpy(), gold["mlpproj.out"], **TOL)
    np.testing.assert_allclose(ores.mlp_proj_model(sd2, torch.zeros_like(e2)).numpy(), gold["mlpproj.zero"], **TOL)


def test_latents_in_half_precision_adapter_dtype(golden_dir):
    """the reference draws and blends in unet.dtype (fp16 in generate.py:77-81): another random sequence than an fp32
    draw, and half-precision rounding inside the blend — bit-exact against the imported reference"""
    gold = _load(golden_dir, "latents_half")
    boxes = [[40 / 512, 150 / 512, 230 / 512, 450 / 512], [280 / 512, 150 / 512, 470 / 512, 450 / 512]]
    for name, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        lst, bg, _ = ol.get_input_latents_list(0, 123456789, 0.01, 512, 512, boxes, dtype=dt)
        assert lst[0].dtype == dt
        assert np.array_equal(bg.float().numpy(), gold[f"{name}.bg"])
        assert np.array_equal(lst[0].float().numpy(), gold[f"{name}.input0"]) and np.array_equal(lst[1].float().numpy(), gold[f"{name}.input1"])
        one = ol.get_input_latents_lne(1, 7, 7 + 123456789, 0.01, 512, 512, boxes, dtype=dt)
        assert np.array_equal(one.float().numpy(), gold[f"{name}.lne_seed7_idx1"])


def test_feed_forward_geglu(golden_dir):
    gold = _load(golden_dir, "ff_geglu")
    sd = {"ff." + k[len("ff.sd."):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("ff.sd.")}
    got = ounet.feed_forward_geglu(sd, "ff", torch.from_numpy(gold["ff.x"]))
    np.testing.assert_allclose(got.numpy(), gold["ff.out"], **TOL)


@pytest.mark.parametrize("nbox", [1, 2, 4])
def test_guidance_losses(golden_dir, nbox):
    gold = _load(golden_dir, "guidance")
    maps, g = gc.guidance_attn_maps(nbox)
    keys = gc.GUIDANCE_KEYS
    for mode, kw in (("max", dict(use_ratio_based_loss=False, fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=4.0)),
                     ("ratio", dict(use_ratio_based_loss=True))):
        saved = {k: v.clone().requires_grad_(True) for k, v in maps.items()}
        loss = og.compute_ca_lossv3(saved, gc.GUIDANCE_BOXES[nbox], gc.GUIDANCE_POSITIONS[nbox], keys, **kw)
        np.testing.assert_allclose(loss.detach().numpy(), gold[f"guid.{nbox}.{mode}.loss"], rtol=1e-6)
        grads = torch.autograd.grad(loss, [saved[k] for k in keys])
        if nbox == 2:
            for k, gr in zip(keys, grads):
                np.testing.assert_allclose(gr.numpy(), gold[f"guid.2.{mode}.grad.{'_'.join(map(str, k))}"], rtol=1e-5, atol=1e-9)
        else:
            np.testing.assert_allclose([float(gr.double().abs().sum()) for gr in grads], gold[f"guid.{nbox}.{mode}.gradsum"], rtol=1e-5)
    if nbox == 2:
        refs = gc.guidance_ref_maps(g)
        loss = og.compute_ca_lossv3(maps, gc.GUIDANCE_BOXES[2], gc.GUIDANCE_POSITIONS[2], keys, ref_ca_saved_attns=refs,
                                    index=3, ref_ca_loss_weight=2.0, word_token_indices=[3, 7],
                                    use_ratio_based_loss=False, fg_top_p=0.2, bg_top_p=0.2)
        np.testing.assert_allclose(loss.numpy(), gold["guid.2.withref.loss"], rtol=1e-6)


def test_phrase_indices(golden_dir):
    gold = _load(golden_dir, "guidance")
    tok = gc.FakeTokenizer()
    prompt = "a photo of a red cat and a small brown dog , park"
    pos, wti = og.get_phrase_indices(tok, prompt, ["a red cat", "a small brown dog"], words=["cat", "dog"],
                                     return_word_token_indices=True)
    assert pos[0] == gold["phrase.pos0"].tolist() and pos[1] == gold["phrase.pos1"].tolist()
    assert wti == gold["phrase.wti"].tolist()
    pos2, newp = og.get_phrase_indices(tok, "a street", ["a blue car"], add_suffix_if_not_found=True)
    assert pos2[0] == gold["phrase.suffix.pos0"].tolist() and newp == str(gold["phrase.suffix.prompt"])


def test_geometry(golden_dir):
    gold = _load(golden_dir, "geometry_latents")
    boxes = gold["geo.boxes"].tolist()
    sp = []
    for b in boxes:
        for (H, Wd) in ((64, 64), (16, 16), (8, 8), (96, 96)):
            sp.append(list(geo.scale_proportion(b, H, Wd)) + list(geo.scale_proportion(b, H, Wd, use_legacy=True)))
    assert np.array_equal(np.array(sp), gold["geo.scale_proportion"])
    assert np.array_equal(np.stack([geo.proportion_to_mask(b, 64, 64).numpy() for b in boxes]), gold["geo.mask64"])
    cen = [geo.get_centered_box(b) for b in boxes[:3]] + [geo.get_centered_box(b, horizontal_center_only=False) for b in boxes[:3]] \
        + [geo.get_centered_box(b, horizontal_center_only=False, vertical_placement="floor_padding", floor_padding=0.05) for b in boxes[:3]]
    np.testing.assert_allclose(np.array(cen), gold["geo.centered"], rtol=0, atol=0)
    masks = [torch.from_numpy(m) for m in gold["geo.masks"]]
    assert np.array_equal(np.array([geo.binary_mask_to_box(m) for m in masks]), gold["geo.mask_box"])
    assert np.array_equal(torch.stack([geo.binary_mask_to_box_mask(m) for m in masks]).numpy(), gold["geo.mask_box_mask"])
    np.testing.assert_allclose(np.array([geo.binary_mask_to_center(m, normalize=True) for m in masks]), gold["geo.mask_center"], rtol=0, atol=0)
    t = torch.randn(3, 1, 4, 64, 64, generator=torch.Generator().manual_seed(901))
    got = np.stack([geo.shift_tensor(t, xo, yo, offset_normalized=True).numpy() for xo, yo in gold["geo.shifts"].tolist()])
    assert np.array_equal(got, gold["geo.shift_out"])
    ts = torch.arange(981, 0, -20)
    assert np.array_equal(geo.get_fast_schedule(ts, 10, 2).numpy(), gold["sched.fast_10_2"])
    assert np.array_equal(geo.get_fast_schedule(ts, 49, 2).numpy(), gold["sched.fast_49_2"])


def test_latents(golden_dir):
    gold = _load(golden_dir, "geometry_latents")
    boxes = gold["geo.boxes"].tolist()
    lst, bg, seeds = ol.get_input_latents_list(0, 123456789, 0.01, 512, 512, boxes[:2])
    assert np.array_equal(lst[0].numpy(), gold["lat.input0"]) and np.array_equal(lst[1].numpy(), gold["lat.input1"])
    assert np.array_equal(bg.numpy(), gold["lat.bg"]) and seeds == gold["lat.seeds"].tolist()
    one = ol.get_input_latents_lne(1, 7, 7 + 123456789, 0.01, 512, 512, boxes[:2])
    assert np.array_equal(one.numpy(), gold["lat.lne_seed7_idx1"])
    g = torch.Generator().manual_seed(900)
    masks = []
    for i in range(3):
        m = torch.zeros(64, 64, dtype=torch.bool)
        y0, x0 = 5 + 11 * i, 8 + 9 * i
        m[y0:y0 + 20 + 3 * i, x0:x0 + 14 + 5 * i] = True
        m &= torch.rand(64, 64, generator=g) > 0.15
        masks.append(m)
    assert np.array_equal(torch.stack(masks).numpy(), gold["geo.masks"])
    lat_all = [torch.randn(51, 1, 4, 64, 64, generator=g) for _ in range(3)]
    new_l, new_m, offs = ol.align_with_bboxes(lat_all, masks, boxes[:3])
    np.testing.assert_allclose(np.array(offs), gold["lat.align_offsets"], rtol=0, atol=0)
    assert np.array_equal(torch.stack(new_m).numpy(), gold["lat.align_masks"])
    cs = [float(x.double().sum()) for x in new_l] + [float(x.double().abs().sum()) for x in new_l]
    np.testing.assert_allclose(cs, gold["lat.align_l_checksum"], rtol=0, atol=0)
    comp, fgidx = ol.compose_latents(new_l, new_m, bg, 51)
    assert np.array_equal(fgidx.numpy(), gold["lat.compose_fgidx"])
    assert np.array_equal(comp[0].numpy(), gold["lat.compose_step0"])
    assert np.array_equal(comp[37].numpy(), gold["lat.compose_step37"])
    np.testing.assert_allclose([float(comp.double().sum()), float(comp.double().abs().sum())], gold["lat.compose_checksum"], rtol=0, atol=0)


def test_clip_vision_oracle_pinned_against_transformers():
    """oracle/clip.py == transformers.CLIPVisionModelWithProjection on the stored tiny models (weights, inputs and outputs
    captured from the installed library by tests/golden/make_clip_golden.py): image_embeds, last and penultimate states."""
    import os
    import numpy as np
    import torch
    from oracle import clip as oc
    from tests.golden import make_clip_golden as mk
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "clip_vision.npz"))
    for name, (hid, inter, layers, heads, img, patch, proj, act) in mk.CASES.items():
        sd = {k[len(name) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(name + ".w.")}
        o = oc.clip_vision_forward(dict(hidden_size=hid, num_attention_heads=heads, patch_size=patch, hidden_act=act), sd,
                                   torch.from_numpy(g[name + ".x"]))
        assert len(o["hidden_states"]) == int(g[name + ".n_hidden_states"]) == layers + 1
        for key, got in (("image_embeds", o["image_embeds"]), ("last_hidden_state", o["last_hidden_state"]),
                         ("penultimate", o["hidden_states"][-2])):
            ref = torch.from_numpy(g[f"{name}.{key}"])
            assert got.shape == ref.shape
            assert float((got - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())), (name, key)


def test_clip_text_oracle_pinned_against_transformers():
    """oracle/clip.py::clip_text_forward == transformers.CLIPTextModel (causal mask, EOS pooling) on the stored tiny models."""
    import os
    import numpy as np
    import torch
    from oracle import clip as oc
    from tests.golden import make_clip_golden as mk
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "clip_text.npz"))
    for name, (vocab, hid, inter, layers, heads, max_pos, act, eos) in mk.TEXT_CASES.items():
        sd = {k[len(name) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(name + ".w.")}
        o = oc.clip_text_forward(dict(hidden_size=hid, num_attention_heads=heads, hidden_act=act, eos_token_id=eos), sd,
                                 torch.from_numpy(g[name + ".ids"]))
        for key in ("last_hidden_state", "pooler_output"):
            ref = torch.from_numpy(g[f"{name}.{key}"])
            assert o[key].shape == ref.shape
            assert float((o[key] - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())), (name, key)


@pytest.mark.parametrize("T", gc.BLOCK_T)
def test_first_level_block_composition(golden_dir, T):
    """round 5: the oracle's BasicTransformerBlock / Transformer2DModel restatement (oracle/unet.py: rows T1 / T2 of SURVEY §8) against the
    composition of the reference's own Attention / processors / FeedForward captured by ``make_golden.py::gen_block`` (block.npz)"""
    gold = _load(golden_dir, "block")
    sd, x, enc = gc.block_params(T)
    p = "t"
    sdp = {p + "." + k: v for k, v in sd.items()}
    got = ounet.transformer_2d(sdp, p, x, enc, gc.BLOCK_HEADS, 1, False, 32, {}, 0.4, T, "ip")
    np.testing.assert_allclose(got.numpy(), gold[f"T{T}.transformer.scale0.4"], rtol=2e-5, atol=5e-6)
    B, C, hh, ww = x.shape
    tok = x.permute(0, 2, 3, 1).reshape(B, hh * ww, C)
    got = ounet.basic_transformer_block(sdp, p + ".transformer_blocks.0", tok, enc, gc.BLOCK_HEADS, {}, 0.4, T, "ip")
    np.testing.assert_allclose(got.numpy(), gold[f"T{T}.block.scale0.4"], rtol=2e-5, atol=5e-6)

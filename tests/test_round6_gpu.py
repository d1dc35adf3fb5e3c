"""GPU, round 6: the ping-pong 256 x 256 GEMM (csrc/tg_gemm_pp.hip, tg_gemm force_tile 24 / planner kind 7) against a plain PyTorch fp32 reference of
the same op, through the C ABI; the drop-in surface closed this round (compose_latents_with_alignment, the reference-signature stage loops)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.test_kernels_gpu import DTYPES, _dev, check, rnd

pytestmark = pytest.mark.gpu

PP_SHAPES = [
    # M, N, K: one tile / one K-tile pair; several tiles per workgroup (the cross-tile prefetch: > 256 tiles); odd K-tile counts; K = 128 (two K-tiles)
    (256, 256, 128), (512, 768, 192), (1024, 512, 64 * 7), (4096, 1280, 1280), (16384, 5120, 320), (4096, 2560, 640), (8192, 1024, 2560),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", PP_SHAPES)
def test_pp_gemm_linear_epilogues(dtype, shape):
    """plain / bias + residual + per-batch vector / activation + scale through the LDS bounce; repeatable bit for bit"""
    from theatergen_amd import ops
    dev = _dev()
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    rows = 128
    a, w = rnd((M, K), dtype, g), rnd((N, K), dtype, g, 1 / math.sqrt(K))
    bias, res, bvec = rnd((N,), dtype, g), rnd((M, N), dtype, g), rnd((M // rows, N), dtype, g)
    ad, wd, bd, rd, vd = a.to(dev), w.to(dev), bias.to(dev), res.to(dev), bvec.to(dev)
    tm, tn, sp, kind = ops.gemm(ad, wd, M, N, K, force_tile=24, plan_only=True)
    assert (tm, tn, sp, kind) == (256, 256, 1, 7)
    ref0 = ad.float() @ wd.float().t()
    check(ops.linear(ad, wd, force_tile=24), ref0.cpu(), dtype, f"pp plain {shape}")
    ref = (ref0 + bd.float() + rd.float() + vd.float().repeat_interleave(rows, 0)).cpu()
    out = ops.linear(ad, wd, bd, res=rd, bvec=vd, rows_per_batch=rows, force_tile=24)
    check(out, ref, dtype, f"pp bias + res + bvec {shape}")
    again = ops.linear(ad, wd, bd, res=rd, bvec=vd, rows_per_batch=rows, force_tile=24)
    same = torch.equal(out, again)
    assert same, "the ping-pong GEMM is deterministic"
    out = ops.linear(ad, wd, bd, act=ops.ACT_SILU, out_scale=0.5, force_tile=24)
    check(out, (F.silu(ref0 + bd.float()) * 0.5).cpu(), dtype, f"pp silu {shape}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_pp_gemm_transposed_sides_are_caught(dtype):
    """asymmetric operands: a swapped operand role or a transposed output tile cannot pass (guide 5.4 rule 16)"""
    from theatergen_amd import ops
    dev = _dev()
    M, N, K = 512, 256, 128
    a = torch.zeros(M, K)
    a[:, 0] = torch.arange(M) % 7 + 1
    w = torch.zeros(N, K)
    w[:, 0] = (torch.arange(N) % 5 + 1) * 0.25
    out = ops.linear(a.to(dev, dtype), w.to(dev, dtype), force_tile=24)
    ref = a.to(dtype).float() @ w.to(dtype).float().t()
    same = torch.equal(out.float().cpu(), ref.to(dtype).float())
    assert same


@pytest.mark.parametrize("dtype", DTYPES)
def test_pp_gemm_qkv_split_geglu_and_layernorm_fold(dtype):
    """the attention operand layout (Q | K token-major + V^T per batch item), the fused GEGLU epilogue and the LayerNorm fold with precomputed row
    statistics (the forms the UNet launches) on the ping-pong tiles vs fp32 references"""
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import pack_geglu, pack_ln_linear
    dev = _dev()
    g = torch.Generator().manual_seed(2606)
    B, rows, C = 2, 1024, 256
    M, N, K = B * rows, 3 * C, C
    x = ((torch.randn(M, C, generator=g) + 0.5 * torch.randn(M, 1, generator=g)) * 2.0).to(dtype)
    gamma, beta = (1 + 0.3 * torch.randn(C, generator=g)).to(dtype), (0.3 * torch.randn(C, generator=g)).to(dtype)
    eps = 1e-5
    xn = F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), eps)
    xd = x.to(dev)
    w3 = rnd((N, K), dtype, g, 1 / math.sqrt(K))
    # (a) plain q | k | v^T
    ref = x.float() @ w3.float().t()
    out = torch.zeros((M, 2 * C), dtype=dtype, device=dev)
    out_t = torch.zeros((B, C, rows), dtype=dtype, device=dev)
    ops.gemm(xd, w3.to(dev), M, N, K, rows_per_batch=rows, out=out, n_split=2 * C, out_t=out_t, ldt=rows, force_tile=24)
    check(out, ref[:, :2 * C], dtype, "pp qkv main")
    check(out_t, ref[:, 2 * C:].reshape(B, rows, C).permute(0, 2, 1), dtype, "pp qkv V^T")
    # (b) the same with the LayerNorm folded in (precomputed statistics)
    st = ops.layernorm_stats(xd, eps)
    wl3, u3, v3 = pack_ln_linear(w3.to(dev), None, gamma.to(dev), beta.to(dev))
    assert ops.gemm(xd, wl3, M, N, K, rows_per_batch=rows, n_split=2 * C, out_t=out_t, ldt=rows, out=out, ln=(u3, v3, eps, st), force_tile=24, plan_only=True)[3] == 7
    out.zero_(); out_t.zero_()
    ops.gemm(xd, wl3, M, N, K, rows_per_batch=rows, out=out, n_split=2 * C, out_t=out_t, ldt=rows, ln=(u3, v3, eps, st), force_tile=24)
    ref3 = xn @ w3.float().t()
    check(out, ref3[:, :2 * C], dtype, "pp ln-folded q|k", scale=1.5)
    check(out_t, ref3[:, 2 * C:].reshape(B, rows, C).permute(0, 2, 1), dtype, "pp ln-folded v^T", scale=1.5)
    # (c) GEGLU with bias, plain and LayerNorm-folded
    N2 = 8 * C
    wf, bf = rnd((N2, C), dtype, g, 1 / math.sqrt(C)), rnd((N2,), dtype, g)
    wp, bp = pack_geglu(wf.to(dev), bf.to(dev))
    y = x.float() @ wf.float().t() + bf.float()
    gg = ops.gemm(xd, wp, M, N2, C, bias=bp, geglu=True, force_tile=24)
    assert gg.shape == (M, N2 // 2)
    check(gg, y[:, :N2 // 2] * F.gelu(y[:, N2 // 2:]), dtype, "pp geglu")
    small = ops.gemm(xd, wp, M, N2, C, bias=bp, geglu=True, force_tile=1)
    check(gg, small.float(), dtype, "pp geglu vs the 128 x 128 kernel", scale=2.0)
    yn = xn @ wf.float().t() + bf.float()
    wlg, ug, vg = pack_ln_linear(wp, bp, gamma.to(dev), beta.to(dev))
    gl = ops.gemm(xd, wlg, M, N2, C, geglu=True, ln=(ug, vg, eps, st), force_tile=24)
    check(gl, yn[:, :N2 // 2] * F.gelu(yn[:, N2 // 2:]), dtype, "pp ln-folded geglu", scale=1.5)


P160_SHAPES = [(256, 160, 128), (512, 480, 192), (1024, 320, 64 * 7), (16384, 640, 640), (4096, 1920, 320), (8192, 640, 2560)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", P160_SHAPES)
def test_pp160_gemm_linear_epilogues(dtype, shape):
    """the ping-pong structure on 256 x 160 tiles (csrc/tg_gemm_pp160.hip, force_tile 25): plain / bias + residual + per-batch vector / activation;
    K of two tiles up to forty (the three-stage ring wraps), one to 768 tiles"""
    from theatergen_amd import ops
    dev = _dev()
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K + 1)
    rows = 128
    a, w = rnd((M, K), dtype, g), rnd((N, K), dtype, g, 1 / math.sqrt(K))
    bias, res, bvec = rnd((N,), dtype, g), rnd((M, N), dtype, g), rnd((M // rows, N), dtype, g)
    ad, wd, bd, rd, vd = a.to(dev), w.to(dev), bias.to(dev), res.to(dev), bvec.to(dev)
    assert ops.gemm(ad, wd, M, N, K, force_tile=25, plan_only=True) == (256, 160, 1, 7)
    ref0 = ad.float() @ wd.float().t()
    check(ops.linear(ad, wd, force_tile=25), ref0.cpu(), dtype, f"pp160 plain {shape}")
    out = ops.linear(ad, wd, bd, res=rd, bvec=vd, rows_per_batch=rows, force_tile=25)
    check(out, (ref0 + bd.float() + rd.float() + vd.float().repeat_interleave(rows, 0)).cpu(), dtype, f"pp160 bias + res + bvec {shape}")
    again = ops.linear(ad, wd, bd, res=rd, bvec=vd, rows_per_batch=rows, force_tile=25)
    same = torch.equal(out, again)
    assert same
    out = ops.linear(ad, wd, bd, act=ops.ACT_SILU, out_scale=0.5, force_tile=25)
    check(out, (F.silu(ref0 + bd.float()) * 0.5).cpu(), dtype, f"pp160 silu {shape}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_pp160_gemm_qkv_split_and_layernorm_fold(dtype):
    """Q | K token-major + V^T per batch item (n_split on an 80-column boundary) and the LayerNorm fold with precomputed row statistics on the 256 x 160
    tiles: the 32 x 32 level's q | k | v projection (C = 640)"""
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import pack_ln_linear
    dev = _dev()
    g = torch.Generator().manual_seed(2607)
    B, rows, C = 2, 512, 320
    M, N, K = B * rows, 3 * C, C
    x = ((torch.randn(M, C, generator=g) + 0.5 * torch.randn(M, 1, generator=g)) * 2.0).to(dtype)
    gamma, beta = (1 + 0.3 * torch.randn(C, generator=g)).to(dtype), (0.3 * torch.randn(C, generator=g)).to(dtype)
    eps = 1e-5
    xn = F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), eps)
    xd = x.to(dev)
    w3 = rnd((N, K), dtype, g, 1 / math.sqrt(K))
    ref = x.float() @ w3.float().t()
    out = torch.zeros((M, 2 * C), dtype=dtype, device=dev)
    out_t = torch.zeros((B, C, rows), dtype=dtype, device=dev)
    ops.gemm(xd, w3.to(dev), M, N, K, rows_per_batch=rows, out=out, n_split=2 * C, out_t=out_t, ldt=rows, force_tile=25)
    check(out, ref[:, :2 * C], dtype, "pp160 qkv main")
    check(out_t, ref[:, 2 * C:].reshape(B, rows, C).permute(0, 2, 1), dtype, "pp160 qkv V^T")
    st = ops.layernorm_stats(xd, eps)
    wl3, u3, v3 = pack_ln_linear(w3.to(dev), None, gamma.to(dev), beta.to(dev))
    out.zero_(); out_t.zero_()
    ops.gemm(xd, wl3, M, N, K, rows_per_batch=rows, out=out, n_split=2 * C, out_t=out_t, ldt=rows, ln=(u3, v3, eps, st), force_tile=25)
    ref3 = xn @ w3.float().t()
    check(out, ref3[:, :2 * C], dtype, "pp160 ln-folded q|k", scale=1.5)
    check(out_t, ref3[:, 2 * C:].reshape(B, rows, C).permute(0, 2, 1), dtype, "pp160 ln-folded v^T", scale=1.5)
    with pytest.raises(RuntimeError):
        ops.linear(rnd((300, K), dtype, g).to(dev), w3.to(dev), force_tile=25)


def test_pp_gemm_batched_a_padded_pitches_and_refusals():
    from theatergen_amd import ops
    dev = _dev()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(77)
    # batched A (a_rows_per_batch / a_batch_stride) and padded row pitches
    B, rows, K, N = 2, 256, 192, 256
    buf = rnd((B, rows + 16, K), dtype, g).to(dev)
    w = rnd((N, K), dtype, g, 1 / math.sqrt(K)).to(dev)
    out = ops.gemm(buf, w, B * rows, N, K, a_rows_per_batch=rows, a_batch_stride=(rows + 16) * K, force_tile=24)
    check(out, (buf[:, :rows].reshape(-1, K).float() @ w.float().t()).cpu(), dtype, "pp batched A")
    wide_a, wide_w = rnd((512, K + 64), dtype, g).to(dev), rnd((N, K + 8), dtype, g, 1 / math.sqrt(K)).to(dev)
    out = ops.linear(wide_a[:, :K], wide_w[:, :K], force_tile=24)
    check(out, (wide_a[:, :K].float() @ wide_w[:, :K].float().t()).cpu(), dtype, "pp padded pitches")
    # not a ping-pong problem: ragged M / N, two-source A -> a loud error under force_tile 24, never another kernel
    with pytest.raises(RuntimeError):
        ops.linear(rnd((300, K), dtype, g).to(dev), w, force_tile=24)
    with pytest.raises(RuntimeError):
        ops.linear(rnd((512, K), dtype, g).to(dev), rnd((320, K), dtype, g).to(dev), force_tile=24)


def test_pp_gemm_is_what_the_planner_picks_for_the_feedforward_shapes():
    from theatergen_amd import ops
    dev = _dev()
    dtype = torch.bfloat16
    for (M, N, K, geglu, want) in [(16384, 5120, 640, True, 7), (4096, 10240, 1280, True, 7), (4096, 1280, 1280, False, None), (16384, 640, 640, False, 7)]:
        a = torch.zeros((M, K), dtype=dtype, device=dev)
        w = torch.zeros((N, K), dtype=dtype, device=dev)
        kind = ops.gemm(a, w, M, N, K, geglu=geglu, plan_only=True)[3]
        if want is not None:
            assert kind == want, (M, N, K, kind)
        else:
            assert kind != 7, (M, N, K, kind)


def test_compose_latents_with_alignment_vs_reference_golden(tmp_path, monkeypatch):
    """the stage-1 -> stage-2 hand-off with the reference's signature (utils/latents.py:242-255 <- theatergen.py:415-423): shift (tg_shift) + pixel paste
    (host) + masked composition (tg_masked_compose) against outputs of the imported reference (tests/golden/mid_image.npz)"""
    import os
    from PIL import Image
    from tests.golden import gen_common as gc
    from theatergen_amd import latents as L
    dev = _dev()
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "mid_image.npz"))
    monkeypatch.chdir(tmp_path)
    masks, images, boxes = gc.mid_image_case(0)
    g = torch.Generator().manual_seed(910)
    masks64 = [m.view(64, 8, 64, 8).any(3).any(1) for m in masks]
    lat_all = [torch.randn(51, 1, 4, 64, 64, generator=g).to(dev) for _ in masks]
    bg = torch.randn(1, 4, 64, 64, generator=g).to(dev)

    class _Cfg:
        in_channels = 4

    class _Unet:
        config = _Cfg()
        dtype = torch.float32

    ad = type("A", (), {"pipe": type("P", (), {"unet": _Unet(), "scheduler": type("Sc", (), {"init_noise_sigma": 1.0})()})()})()
    comp, fgidx, inp_mask, inp_img = L.compose_latents_with_alignment(
        "1.5", ad, 0, masks, [Image.fromarray(i) for i in images], None, lat_all, masks64, 50, 1, 512, 512,
        align_with_overall_bboxes=True, overall_bboxes=[[boxes[0]], [boxes[1]]], horizontal_shift_only=False, latents_bg=bg)
    assert np.array_equal(fgidx.cpu().numpy(), gold["cwa.fgidx"])
    assert torch.equal(comp[0].cpu(), torch.from_numpy(gold["cwa.step0"])) and torch.equal(comp[23].cpu(), torch.from_numpy(gold["cwa.step23"]))
    np.testing.assert_allclose([float(comp.double().sum()), float(comp.double().abs().sum())], gold["cwa.checksum"], rtol=1e-12)
    assert np.array_equal(np.array(inp_mask), gold["cwa.mask"]) and np.array_equal(np.array(inp_img), gold["cwa.image"])
    # the reference has no value to return when nothing is aligned (utils/latents.py:250-255): the caller's skip-the-turn policy sees a RuntimeError
    with pytest.raises(RuntimeError):
        L.compose_latents_with_alignment("1.5", ad, 0, masks, images, None, lat_all, masks64, 50, 1, 512, 512, align_with_overall_bboxes=False,
                                         overall_bboxes=[[boxes[0]], [boxes[1]]], latents_bg=bg)


# ---- the reference's two stage functions with their own signatures, on the engine ------------------------------------------------
class _FakeTextPipe:
    """what ``generate_semantic_guidance`` / ``final_image_generation`` read from ``adapter.pipe`` besides the UNet: ``encode_prompt`` (seeded by the
    prompt strings, records them), ``vae``, ``scheduler``, ``vae_scale_factor``"""

    def __init__(self, unet, vae, ctx):
        from theatergen_amd.scheduler import DDIMScheduler
        self.unet, self.vae, self.ctx = unet, vae, ctx
        self.scheduler = DDIMScheduler()
        self.vae_scale_factor = 8
        self.controlnet = None
        self.prompts = []

    def encode_prompt(self, prompt, device=None, num_images_per_prompt=1, do_classifier_free_guidance=True, negative_prompt=None, **kw):
        self.prompts.append((prompt, negative_prompt))

        def emb(s):
            g = torch.Generator().manual_seed(sum(map(ord, s)) % 100003)
            return (torch.randn(1, 77, self.ctx, generator=g) * 0.5).to(device=self.unet.device, dtype=self.unet.dtype)
        return emb(prompt), emb(negative_prompt)


def _image_tokens_of(pil_image, ctx, dev, dtype):
    """stand-in for CLIP + image projection: tokens seeded by the image's pixels (cond) / fixed (uncond)"""
    seed = int(np.asarray(pil_image.convert("RGB").resize((8, 8))).astype(np.int64).sum() % 100003)
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(1, 4, ctx, generator=g) * 0.5).to(dev, dtype), (torch.randn(1, 4, ctx, generator=torch.Generator().manual_seed(5)) * 0.5).to(dev, dtype)


def test_generate_semantic_guidance_reference_signature_vs_oracle_loop(tmp_path, monkeypatch):
    """theatergen.py:108-135 calls ``pipelines.generate_semantic_guidance(task, fg_seed_now, basever, ip_prompt, database_path, idx, adapter, model_dict,
    input_latents, input_embeddings, num_inference_steps, bboxes, phrases, object_positions, guidance_scale=..., return_saved_cross_attn=True,
    return_box_vis=True, save_all_latents=True, ...)`` and unpacks five values: the same call on the engine, against the free-running oracle loop —
    first appearance of a character (placeholder image, IP scale 0, PNG written), its reuse (PNG read, scale 0.4), the fast schedule."""
    from PIL import Image
    from oracle import ddim as oddim
    from oracle import unet as ou
    from oracle import vae as ov
    from tests.test_hotpath_gpu import DEV, _build, _build_vae, close, net_tol
    from theatergen_amd import config, pipelines
    from theatergen_amd.ip_adapter import IPAdapter
    from theatergen_amd.schedule import get_fast_schedule
    from theatergen_amd.vae import tiny_vae_config
    dtype = torch.bfloat16
    cfg = config.tiny()
    unet, sd_r = _build(cfg, dtype)
    vcfg = tiny_vae_config()
    vae, vsd = _build_vae(vcfg, dtype)
    pipe = _FakeTextPipe(unet, vae, cfg.cross_attention_dim)
    ad = IPAdapter(pipe, None, None, DEV, num_tokens=4)
    monkeypatch.setattr(ad, "get_image_embeds", lambda pil_image=None, clip_image_embeds=None: _image_tokens_of(pil_image, cfg.cross_attention_dim, DEV, dtype))
    monkeypatch.chdir(tmp_path)
    Image.fromarray(np.full((32, 32, 3), 90, np.uint8)).save("model.png")
    db = str(tmp_path) + "/db_"
    g = torch.Generator().manual_seed(31)
    steps = 4
    lat = torch.randn(1, 4, 16, 16, generator=g).to(dtype)

    def oracle(enc, scale, timesteps=None, n=steps):
        osch = oddim.DDIMSchedule()
        osch.set_timesteps(n)
        ref, rows = lat.float().clone(), [lat.float().clone()]
        for t in (osch.timesteps if timesteps is None else timesteps).tolist():
            npred = ou.unet_forward(cfg, sd_r, torch.cat([ref] * 2).to(dtype).float(), t, enc.float().cpu(), ip_scale=scale)
            ref = oddim.step_epilogue(osch, npred, t, ref, 7.5)
            rows.append(ref.clone())
        return ref, torch.stack(rows)

    def enc_of(prompt, pil):
        pos, neg = pipe.encode_prompt(prompt, negative_prompt=pipelines.SINGLE_OBJECT_NEGATIVE_PROMPT)
        img, unc = _image_tokens_of(pil, cfg.cross_attention_dim, DEV, dtype)
        return torch.cat([torch.cat([neg, unc], 1), torch.cat([pos, img], 1)], 0)

    kw = dict(guidance_scale=7.5, return_cross_attn=False, return_saved_cross_attn=True, semantic_guidance_kwargs=None, saved_cross_attn_keys=[("down", 2, 1, 0)],
              return_cond_ca_only=True, return_token_ca_only=3, offload_cross_attn_to_cpu=False, return_box_vis=True, save_all_latents=True,
              dynamic_num_inference_steps=True, use_adapter=True, obj_id=7)
    # (1) first appearance: no db_7.png -> placeholder, scale 0, the decoded image becomes the character's reference
    out = pipelines.generate_semantic_guidance("story", 123, "1.5", "a red fox", db, 0, ad, None, lat.to(DEV), None, steps, None, None, None, **kw)
    assert len(out) == 5
    latents, image, saved, image2, latents_all = out
    assert pipe.prompts[-1] == ("full-body picture of a red fox", pipelines.SINGLE_OBJECT_NEGATIVE_PROMPT)
    assert saved == [{}] * steps and image2 is image and image.size == (128, 128)
    assert latents_all.shape == (steps + 1, 1, 4, 16, 16) and latents_all.device.type == "cpu" and latents_all.dtype == dtype and latents.dtype == dtype
    assert os.path.exists(db + "7.png"), "the first image of a character is written to the database (models/pipelines.py:476-477)"
    ref, rows = oracle(enc_of("full-body picture of a red fox", Image.open("model.png")), 0.0)
    close(latents, ref, net_tol(dtype), "generate_semantic_guidance: first appearance (scale 0)")
    close(latents_all, rows, net_tol(dtype), "generate_semantic_guidance: latents_all")
    dec = ov.decode(vcfg, vsd, latents.float().cpu())
    got_img = torch.from_numpy(np.asarray(image).astype(np.float32) / 255.0).permute(2, 0, 1)[None]
    close(got_img, (dec / 2 + 0.5).clamp(0, 1), 8e-2, "generate_semantic_guidance: decoded image")
    # (2) reuse: db_7.png exists -> its tokens at scale 0.4; editing prompt; latents stay on the device when asked
    out2 = pipelines.generate_semantic_guidance("editing", 123, "1.5", "a red fox", db, 0, ad, None, lat.to(DEV), None, steps, None, None, None,
                                                **{**kw, "return_saved_cross_attn": False, "return_box_vis": False, "offload_latents_to_cpu": False})
    assert len(out2) == 3 and out2[2].device.type == "cuda"
    assert pipe.prompts[-1][0] == "single object, a red fox"
    ref2, _ = oracle(enc_of("single object, a red fox", Image.open(db + "7.png")), 0.4)
    close(out2[0], ref2, net_tol(dtype), "generate_semantic_guidance: reuse (scale 0.4)")
    differs = not torch.equal(out2[0], latents)
    assert differs
    # (3) the fast schedule (utils/schedule.py:4-8 at models/pipelines.py:383-384)
    osch = oddim.DDIMSchedule()
    osch.set_timesteps(6)
    fast = get_fast_schedule(osch.timesteps, 2, 2)
    out3 = pipelines.generate_semantic_guidance("story", 123, "1.5", "a red fox", db, 0, ad, None, lat.to(DEV), None, 6, None, None, None,
                                                **{**kw, "fast_after_steps": 2, "fast_rate": 2})
    assert out3[4].shape[0] == len(fast) + 1
    ref3, _ = oracle(enc_of("full-body picture of a red fox", Image.open(db + "7.png")), 0.4, timesteps=fast, n=6)
    close(out3[0], ref3, net_tol(dtype), "generate_semantic_guidance: fast schedule")
    # refusals
    with pytest.raises(NotImplementedError):
        pipelines.generate_semantic_guidance("story", 1, "xl", "x", db, 0, ad, None, lat.to(DEV), None, steps, None, None, None, obj_id=7)
    with pytest.raises(RuntimeError):
        pipelines.generate_semantic_guidance("story", 1, "1.5", "x", db, 0, ad, None, lat.to(DEV), None, steps, None, None, None)


def test_final_image_generation_reference_signature_vs_oracle_loop(tmp_path, monkeypatch):
    """theatergen.py:448-484 calls ``pipelines.final_image_generation(basever, processor, controlnetpipe, tpipe, overall_prompt, overall_negative_prompt,
    bg_prompt, single_obj_img_list, objects, repeat_ind, height, width, bg_seed, inp_mask, inp_img, adapter, model_dict, composed_latents, frozen_mask,
    bg_input_embeddings, overall_input_embeddings, num_inference_steps, frozen_steps, ...)``: same call on the engine — the frozen latents it writes into
    ``latents_all`` (VAE encoder + re-noising from the device generator), the mask from ``inp_mask``, then ControlNet + UNet + frozen-mask loop vs the oracle."""
    from PIL import Image
    from oracle import controlnet as oc
    from oracle import ddim as oddim
    from oracle import unet as ou
    from oracle import vae as ov
    from tests.test_hotpath_gpu import DEV, _build, _build_controlnet, _build_vae_full, close, net_tol
    from theatergen_amd import config, pipelines
    from theatergen_amd.ip_adapter import IPAdapter
    from theatergen_amd.vae import tiny_vae_config
    dtype = torch.bfloat16
    cfg = config.tiny()
    unet, sd_u = _build(cfg, dtype)
    net, sd_c = _build_controlnet(cfg, dtype)
    vcfg = tiny_vae_config()
    vae, vsd = _build_vae_full(vcfg, dtype)
    pipe = _FakeTextPipe(unet, vae, cfg.cross_attention_dim)
    ad = IPAdapter(pipe, None, None, DEV, num_tokens=4)
    monkeypatch.setattr(ad, "get_image_embeds", lambda pil_image=None, clip_image_embeds=None: _image_tokens_of(pil_image, cfg.cross_attention_dim, DEV, dtype))
    cnpipe = type("CNPipe", (), {"controlnet": net})()
    H = W = 128
    steps, frozen_steps, bg_seed = 4, 3, 11
    g = torch.Generator().manual_seed(41)
    pasted = Image.fromarray((torch.rand(H, W, 3, generator=g) * 255).to(torch.uint8).numpy())
    m512 = np.full((H, W), 255, np.uint8)
    m512[40:100, 24:80] = 0                                   # a character was pasted here
    inp_mask = Image.fromarray(m512, mode="L")
    char_img = Image.fromarray(np.full((32, 32, 3), 140, np.uint8))
    text = (torch.randn(2, 77, cfg.cross_attention_dim, generator=g) * 0.5).to(DEV, dtype)
    latents_all = torch.zeros(steps + 1, 1, 4, H // 8, W // 8, device=DEV)
    calls = []

    def processor(arr):                                       # the lineart detector's place: ndarray in, PIL out
        calls.append(arr.shape)
        return Image.fromarray(255 - arr)
    latents, images = pipelines.final_image_generation("1.5", processor, cnpipe, 1, "two foxes in a park", "lowres", "a park", [char_img], None, 0, H, W,
                                                       bg_seed, inp_mask, pasted, ad, None, latents_all, torch.ones(16, 16), None, (text, text[:1], text[1:]),
                                                       steps, frozen_steps, guidance_scale=7.5)
    assert calls == [(H, W, 3)] and images.shape == (1, H, W, 3) and images.dtype == np.uint8 and latents.shape == (1, 4, 16, 16)
    assert pipe.prompts[-1] == ("two foxes in a park", "lowres")
    assert float(ad.pipe.unet.attn_processors[[k for k in ad.pipe.unet.attn_processors if "attn2" in k][0]].scale) == pytest.approx(0.1)
    # the device generator's draws, in the reference's order (:625-632): posterior noise, re-noising noise, background latents
    g2 = torch.Generator(DEV).manual_seed(bg_seed)
    n1 = torch.randn((1, 4, 16, 16), generator=g2, device=DEV, dtype=dtype)
    n2 = torch.randn((1, 4, 16, 16), generator=g2, device=DEV, dtype=dtype)
    bg = torch.randn((1, 4, 16, 16), generator=g2, device=DEV, dtype=dtype)
    same = torch.equal(latents_all[0], bg.float())
    assert same, "latents_all[0] = fresh background noise, third draw of the bg_seed generator"
    img = (2.0 * torch.from_numpy(np.asarray(pasted).astype(np.float32) / 255.0)[None].permute(0, 3, 1, 2) - 1.0).to(dtype).float()
    mom = ov.encode_moments(vcfg, vsd, img)
    mean, logvar = mom[:, :4], mom[:, 4:].clamp(-30, 20)
    init = vcfg.scaling_factor * (mean + torch.exp(0.5 * logvar) * n1.float().cpu())
    osch = oddim.DDIMSchedule()
    osch.set_timesteps(steps)
    want_rows = torch.stack([osch.add_noise(init, n2.float().cpu(), t) for t in osch.timesteps.tolist()])
    close(latents_all[1:].cpu(), want_rows.reshape(steps, 1, 4, 16, 16), net_tol(dtype), "final_image_generation: frozen latents = VAE-encoded pasted image re-noised at every timestep")
    # the loop: the oracle starts from what the function wrote (its own frozen rows) so that only the loop is compared
    mask = torch.from_numpy(1 - (np.array(inp_mask.resize((16, 16)).convert("L")).astype(np.float32) / 255.0 > 0).astype(np.float32))
    ip_rows = torch.cat([torch.cat([pipe.encode_prompt("two foxes in a park", negative_prompt="lowres")[1], _image_tokens_of(char_img, cfg.cross_attention_dim, DEV, dtype)[1]], 1),
                         torch.cat([pipe.encode_prompt("two foxes in a park", negative_prompt="lowres")[0], _image_tokens_of(char_img, cfg.cross_attention_dim, DEV, dtype)[0]], 1)], 0)
    cond = torch.from_numpy(np.asarray(Image.fromarray(255 - np.asarray(pasted)).resize((W, H), resample=Image.LANCZOS)).astype(np.float32) / 255.0).permute(2, 0, 1)[None]
    cond = torch.cat([cond] * 2).to(dtype).float()
    frozen = latents_all.cpu()
    ref = frozen[0].clone()
    for i, t in enumerate(osch.timesteps.tolist()):
        mi = torch.cat([ref] * 2).to(dtype).float()
        rd, rm = oc.controlnet_forward(cfg, sd_c, mi, t, text.float().cpu(), cond, 1.0, cross_mode="cn")
        rd = [d.to(dtype).float() for d in rd]
        npred = ou.unet_forward(cfg, sd_u, mi, t, ip_rows.float().cpu(), ip_scale=0.1, down_block_additional_residuals=rd, mid_block_additional_residual=rm.to(dtype).float())
        ref = oddim.step_epilogue(osch, npred, t, ref, 7.5, frozen[i + 1] if i < frozen_steps else None, mask)
    close(latents, ref, net_tol(dtype), "final_image_generation: ControlNet + UNet + frozen-mask loop")
    dec = ov.decode(vcfg, vsd, latents.float().cpu(), scaling_factor=0.18215)
    close(torch.from_numpy(images.astype(np.float32) / 255.0).permute(0, 3, 1, 2), (dec / 2 + 0.5).clamp(0, 1), 8e-2, "final_image_generation: decoded image")
    with pytest.raises(TypeError):
        pipelines.final_image_generation("1.5", processor, type("P", (), {"controlnet": object()})(), 1, "p", "n", "b", [char_img], None, 0, H, W, bg_seed, inp_mask,
                                         pasted, ad, None, latents_all, None, None, (text, text[:1], text[1:]), steps, frozen_steps)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("geom", [(8, 64, 64, 320, 320), (16, 32, 32, 640, 640), (16, 32, 32, 320, 640), (2, 64, 64, 640, 1280)])
def test_groupnorm_partials_from_the_slab_conv_epilogue(dtype, geom):
    """round 6: the two-wave slab conv writes the GroupNorm(32) partial sums of its OUTPUT (tg_gemm_desc.out_gn_partials) and tg_groupnorm_from_partials folds
    them: coefficients and the normalised tensor against tg_groupnorm_coef / tg_groupnorm of the same stored output (statistics agree to fp32 rounding) and
    against F.group_norm in fp32; the conv output itself is unchanged bit for bit; a kernel that cannot write them leaves gn_out alone / refuses the pointer."""
    from theatergen_amd import ops
    from theatergen_amd._lib import GemmDesc
    from theatergen_amd.weights_pack import pack_conv3x3
    dev = _dev()
    B, h, w, cin, cout = geom
    g = torch.Generator().manual_seed(B * h + cin + cout)
    x = rnd((B * h * w, cin), dtype, g).to(dev)
    wt = rnd((cout, cin, 3, 3), dtype, g, 1 / math.sqrt(9 * cin))
    bias = rnd((cout,), dtype, g).to(dev)
    res = (rnd((B * h * w, cout), dtype, g) + 0.5).to(dev)                 # a mean the E[x^2] - mean^2 form has to cancel
    bvec = rnd((B, cout), dtype, g).to(dev)
    gamma, beta = (1 + 0.2 * torch.randn(cout, generator=g)).to(dtype).to(dev), (0.2 * torch.randn(cout, generator=g)).to(dtype).to(dev)
    wp = pack_conv3x3(wt).to(dev)
    kw = dict(bias=bias, res=res, bvec=bvec, rows_per_batch=h * w)
    plain = ops.conv3x3(x, wp, B, h, w, cin, **kw)
    gn = {"groups": 32}
    out = ops.conv3x3(x, wp, B, h, w, cin, gn_out=gn, **kw)
    assert "partials" in gn and gn["nblk"] == h * w // 64, gn.keys()
    assert torch.equal(out, plain)
    # the sums themselves: fp64 sums of the stored values per (image, 64-pixel block, group)
    o64 = out.double().reshape(B, h * w // 64, 64, 32, cout // 32)
    want = torch.stack([o64.sum(dim=(2, 4)), (o64 * o64).sum(dim=(2, 4))], dim=-1)
    got = gn["partials"].double()
    assert torch.allclose(got, want, rtol=2e-5, atol=2e-3), (got - want).abs().max().item()
    for eps in (1e-5, 1e-6):
        c_ref = ops.groupnorm_coef(out, B, h * w, 32, eps, gamma, beta)
        c_got = ops.groupnorm_from_partials(gn, B, h * w, cout, eps, gamma, beta)
        assert torch.allclose(c_got, c_ref, rtol=2e-5, atol=2e-6), (c_got - c_ref).abs().max().item()
    for silu in (False, True):
        y_ref = ops.groupnorm(out, B, h * w, 32, 1e-6, gamma, beta, silu=silu)
        y_got = ops.groupnorm_from_partials(gn, B, h * w, cout, 1e-6, gamma, beta, x=out, silu=silu)
        t = F.group_norm(out.float().reshape(B, h * w, cout).transpose(1, 2), 32, gamma.float(), beta.float(), 1e-6)
        t = (F.silu(t) if silu else t).transpose(1, 2).reshape(B * h * w, cout)
        check(y_got, t.cpu(), dtype, f"groupnorm_from_partials {geom} silu={silu}")
        ulp = (y_got.float() - y_ref.float()).abs().max().item()
        assert ulp <= (0.04 if dtype == torch.bfloat16 else 0.006), ulp     # the same apply pass on statistics that agree to ~1e-6: rare 1-ulp flips
    # the one-wave slab kernel and a K-split launch do not write them: gn_out is left alone, the raw pointer is refused
    for extra in (dict(force_tile=12), ):
        gn2 = {"groups": 32}
        ops.conv3x3(x, wp, B, h, w, cin, gn_out=gn2, **extra, **kw)
        assert "partials" not in gn2
    gn3 = {"groups": 24}                                                   # 80 % (cout / 24) != 0 for these widths, or cout % 24 != 0
    ops.conv3x3(x, wp, B, h, w, cin, gn_out=gn3, **kw)
    assert "partials" not in gn3


def test_groupnorm_partials_env_switch_and_unet_block_parity(monkeypatch):
    """ResnetBlock2D -> Transformer2DModel.norm with the statistics from the conv epilogues (default) against the statistics launches (TG_GN_EPI=0)."""
    import importlib
    from theatergen_amd import ops, unet as unet_mod
    dev = _dev()
    dtype = torch.bfloat16
    torch.manual_seed(3)
    blk = unet_mod.ResnetBlock2D(320, 320, 1280).to(dev).to(dtype)
    with torch.no_grad():
        for p in blk.parameters():
            p.copy_(torch.randn_like(p) * (0.05 if p.ndim > 1 else 0.3) + (1.0 if p.ndim == 1 and p.numel() == 320 else 0.0))
    blk.temb_slot = (0, 320)
    B, h, w = 8, 64, 64
    x = unet_mod._Act(torch.randn(B * h * w, 320, device=dev).to(dtype), B, h, w, 320)
    tproj = torch.randn(B, 320, device=dev).to(dtype)
    launches = []
    real = ops.groupnorm_coef
    monkeypatch.setattr(ops, "groupnorm_coef", lambda *a, **k: (launches.append(1), real(*a, **k))[1])
    monkeypatch.setattr(unet_mod, "_GN_EPI", True)
    y1 = blk.run(x, None, tproj)
    n_on = len(launches)
    assert y1.gn is not None and "partials" in y1.gn
    monkeypatch.setattr(unet_mod, "_GN_EPI", False)
    y0 = blk.run(x, None, tproj)
    assert y0.gn is None and len(launches) - n_on == 2 and n_on == 1        # norm1 keeps its statistics launch (its input is not a conv output here)
    d = (y1.t.float() - y0.t.float()).abs().max().item()
    assert d <= 0.07 * y0.t.float().abs().max().item(), d
    rel = ((y1.t.float() - y0.t.float()).norm() / y0.t.float().norm()).item()
    assert rel < 2e-3, rel

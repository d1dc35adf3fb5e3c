"""GPU parity at the sizes, batch and step counts the benchmark runs (VERDICT r1 "What's weak" 1-3):

  * the full SD-1.5 UNet at 512x512 with the bench's CFG batch 16 (M = 65 536 token rows: the tile / tail-split / halo
    heuristics pick other paths than at batch 2) against the fp32 oracle for the rows of two of the eight images, and
    against the HIP result of the same rows computed as a batch-2 call;
  * BASELINE.json configs[0]: SD-1.5 512x512, ONE character box, 20 DDIM steps, latents by the reference recipe, HIP
    (bf16, hipGraph engine) against the fp32 oracle loop, error recorded at steps 1 / 5 / 10 / 20;
  * the boundary driven the way the reference drives it (models/pipelines.py:406-453): per step ``torch.cat([latents] * 2)``
    -> ``unet(x, t, encoder_hidden_states=ip_embeds, cross_attention_kwargs=None, return_dict=False)[0]`` -> CFG ->
    ``scheduler.step(...).prev_sample``, two characters back to back, each with a FRESH ``torch.cat`` of its embeddings that
    is deleted before the next one is built (the caching allocator hands the second tensor the first one's address).

Metrics (tests/parity_metrics.py): relative L2 AND max|err| / max|ref|; tolerances are written next to each comparison and
collected in DESIGN.md section 4 with the measured values.
"""
import os

import pytest
import torch

from tests import parity_metrics as pm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _threads():
    torch.set_num_threads(min(64, os.cpu_count() or 8))


def _build(cfg, dtype, T=4, scale=0.4, seed=0):
    from theatergen_amd import weights as W
    from theatergen_amd.unet import UNet2DConditionModel
    sd = W.random_unet_state_dict(cfg, seed=seed)
    sd_r = {k: v.to(dtype).float() for k, v in sd.items()}          # the oracle sees the storage-rounded weights
    unet = UNet2DConditionModel.from_state_dict(cfg, sd, device=DEV, dtype=dtype, num_tokens=T, ip_scale=scale)
    return unet, sd_r


def test_unet_sd15_cfg_batch16_vs_oracle_and_vs_batch2():
    """bench workload shape: 8 character images -> CFG batch 16 (rows 0-7 negative, 8-15 positive prompts)."""
    from oracle import unet as ou
    from theatergen_amd import config
    dtype = torch.bfloat16
    cfg = config.sd15()
    unet, sd_r = _build(cfg, dtype)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(16, 4, 64, 64, generator=g)
    enc = torch.randn(16, 81, 768, generator=g) * 0.5
    t = 741
    with torch.no_grad():
        out16 = unet(x.to(DEV, dtype), t, enc.to(DEV, dtype), out_dtype=torch.float32).sample.cpu()
    _threads()
    for img in (1, 6):                                     # image `img` = rows img (uncond) and 8 + img (cond)
        rows = [img, 8 + img]
        ref = ou.unet_forward(cfg, sd_r, x[rows].to(dtype).float(), t, enc[rows].to(dtype).float(), ip_scale=0.4, num_tokens=4)
        # whole-net bf16 (~60 layers of bf16 activations, fp32 accumulation): rel-L2 <= 1.0e-2, max <= 3e-2 of the peak
        pm.check(out16[rows], ref, f"sd15 512^2 CFG batch 16, image {img} vs fp32 oracle", 1.0e-2, 3.0e-2)
        with torch.no_grad():
            out2 = unet(x[rows].to(DEV, dtype), t, enc[rows].to(DEV, dtype), out_dtype=torch.float32).sample.cpu()
        # same kernels, other tile / split plans: only fp32 summation order and bf16 re-rounding differ
        # (measured 8.6e-3 / 8.9e-3: two bf16 evaluations differ from each other about as much as each from the fp32 oracle)
        pm.check(out16[rows], out2, f"sd15 512^2 batch-16 rows of image {img} vs the same rows as a batch-2 call", 1.2e-2, 3.0e-2)


def test_config1_sd15_one_box_20_steps_vs_oracle_loop():
    """BASELINE.json configs[0]: the 20-step, 1-box workload end to end: ~100 s of fp32 CPU oracle at 64 threads."""
    from oracle import ddim as oddim
    from oracle import unet as ou
    from theatergen_amd import config, latents as L, story
    from theatergen_amd.ip_adapter import IPAdapter
    from theatergen_amd.pipelines import DenoiseEngine, SDPipe
    dtype = torch.bfloat16
    cfg = config.sd15()
    unet, sd_r = _build(cfg, dtype)
    adapter = IPAdapter(SDPipe(unet), None, None, DEV, num_tokens=4)
    adapter.set_scale(0.4)
    steps = 20
    # latents: the reference recipe (utils/latents.py:257-295) with generate.py's seeds, one character box
    lat_list, _, _ = L.get_input_latents_list(None, bg_seed=0, fg_seed_start=123456789, fg_blending_ratio=0.01, height=512, width=512,
                                              adapter=adapter, so_boxes=[story.box_xyxy(0)])
    lat = lat_list[0].float().cpu()
    g = torch.Generator().manual_seed(77)
    enc = torch.randn(2, 81, 768, generator=g) * 0.5
    eng = DenoiseEngine(unet, None, n_img=1, height=512, width=512, num_inference_steps=steps, guidance_scale=7.5, enc_len=81)
    eng.set_conditioning(enc.to(DEV, dtype))
    hist = eng.run(lat).cpu()
    assert hist.shape == (steps + 1, 1, 4, 64, 64) and torch.equal(hist[0], lat)
    _threads()
    osch = oddim.DDIMSchedule()
    osch.set_timesteps(steps)
    ref = lat.clone()
    encr = enc.to(dtype).float()
    curve = {}
    for i, t in enumerate(osch.timesteps.tolist()):
        mi = torch.cat([ref] * 2).to(dtype).float()            # `.half()` of pipelines.py:414 in the storage dtype
        ref = oddim.step_epilogue(osch, ou.unet_forward(cfg, sd_r, mi, t, encr, ip_scale=0.4, num_tokens=4), t, ref, 7.5)
        if i + 1 in (1, 5, 10, 20):
            curve[i + 1] = pm.metrics(hist[i + 1], ref)
            pm.record(f"config1 sd15 512^2 1 box: latents after step {i + 1}/20 vs fp32 oracle loop", curve[i + 1], step=i + 1)
    # measured (profiles/r2_parity_drift.json): rel-L2 6.1e-3 after step 1, 7.8e-3 after steps 5 / 10 / 20 — the bf16 error
    # of one CFG UNet call, NOT compounding over the chain.  Tolerances: every recorded step rel-L2 <= 1.5e-2, max <= 3e-2.
    for k, m in curve.items():
        assert m["finite"] and m["rel_l2"] <= 1.5e-2 and m["max_rel"] <= 3.0e-2, f"latents after step {k}/20: {m}"


def test_bench_shape_8_images_50_steps_vs_oracle_loop():
    """BASELINE.json configs[1] at the shape bench.py times: ONE whole story = 8 character images, CFG batch 16,
    50 DDIM steps on the hipGraph engine, built by the same story.* helpers as bench.py; the fp32 CPU oracle loop follows ONE of the
    eight images (image 5 = turn 3, character 1) for all 50 steps (~4.5 min at 64 threads).
    Asserted at steps 1 / 10 / 25 / 50: latents rel-L2 <= 1.5e-2, max <= 3e-2; and the network's own contribution (x0 recovered in
    fp64 from consecutive history rows, tests/parity_metrics.py::ddim_net_terms) does NOT compound: rel-L2 of x0 at step 50 within
    1.5x of its value at step 1, every mark <= 4e-2 (measured round 3 at batch 1: 2.3e-2 -> 1.3e-2 -> 1.1e-2)."""
    from theatergen_amd import config, story
    from theatergen_amd.ip_adapter import IPAdapter
    from theatergen_amd.pipelines import DenoiseEngine, SDPipe
    dtype = torch.bfloat16
    cfg = config.sd15()
    unet, sd_r = _build(cfg, dtype)
    adapter = IPAdapter(SDPipe(unet), None, None, DEV, num_tokens=4)
    adapter.set_scale(0.4)
    steps, T, ctx = 50, 4, cfg.cross_attention_dim
    jobs = story.story_jobs(3)                                  # dialogue 3: 4 turns x 2 characters
    assert len(jobs) == 8
    shared = story.shared_conditioning(ctx, T, dtype, DEV)
    char_ids = sorted({j.char_id for j in jobs})
    img_tok = story.character_image_tokens(char_ids, ctx, T, dtype, DEV)
    cidx = {c: i for i, c in enumerate(char_ids)}
    enc = story.job_conditioning(jobs, shared, img_tok, cidx, ctx, dtype, DEV)          # [16, 81, 768]: negatives first
    lat = story.job_latents(jobs, adapter)                                              # [8, 4, 64, 64] fp32
    eng = DenoiseEngine(unet, None, n_img=8, height=512, width=512, num_inference_steps=steps, guidance_scale=7.5, enc_len=77 + T)
    eng.set_conditioning(enc)
    hist = eng.run(lat).cpu()
    assert hist.shape == (steps + 1, 8, 4, 64, 64) and torch.isfinite(hist).all()
    _threads()
    img = 5
    curve = pm.oracle_chain_metrics(cfg, sd_r, hist[:, img:img + 1], lat[img:img + 1].cpu(), enc[[img, 8 + img]].float().cpu(), dtype, steps,
                                    marks=(1, 10, 25, 50),
                                    log=lambda k, m: pm.record(f"bench shape (8 images, CFG batch 16): image {img} after step {k}/50 vs fp32 oracle loop",
                                                               m, step=k))
    for k, m in curve.items():
        assert m["finite"] and m["rel_l2"] <= 1.5e-2 and m["max_rel"] <= 3.0e-2, f"latents after step {k}/50: {m}"
        assert m["x0"]["rel_l2"] <= 4e-2, f"x0 at step {k}/50: {m['x0']}"
    assert curve[50]["x0"]["rel_l2"] <= 1.5 * curve[1]["x0"]["rel_l2"], f"x0 error compounds: {curve[1]['x0']} -> {curve[50]['x0']}"


def test_reference_shaped_loop_two_characters_fresh_embeddings():
    """reference models/pipelines.py:406-453 on the drop-in boundary (INTEGRATION.md level 1), two characters back to back."""
    from oracle import ddim as oddim
    from oracle import unet as ou
    from theatergen_amd import config
    from theatergen_amd.pipelines import prepare_ip_embeds
    from theatergen_amd.scheduler import DDIMScheduler
    dtype = torch.float16                                   # the reference's GPU dtype (generate.py:77-81)
    cfg = config.tiny()
    unet, sd_r = _build(cfg, dtype)
    scheduler = DDIMScheduler()
    steps, gs = 4, 7.5
    scheduler.set_timesteps(steps)
    g = torch.Generator().manual_seed(5)
    D = cfg.cross_attention_dim
    chars = []
    for c in range(2):
        chars.append(dict(lat=torch.randn(1, 4, 16, 16, generator=g), pos=torch.randn(1, 77, D, generator=g) * 0.5,
                          neg=torch.randn(1, 77, D, generator=g) * 0.5, img=torch.randn(1, 4, D, generator=g) * 0.5,
                          uimg=torch.randn(1, 4, D, generator=g) * 0.5))
    osch = oddim.DDIMSchedule()
    osch.set_timesteps(steps)
    ptrs, results = [], []
    # the final cat of each character's embeddings is allocated from a private pool that never holds anything else: when
    # character 0's tensor is deleted its block is the only free one there, so character 1's cat MUST land on it
    pool = torch.cuda.MemPool()
    for ch in chars:
        # prepare_ip_embeds (pipelines.py:860-950): a FRESH cat per character
        parts = (ch["pos"].to(DEV, dtype), ch["neg"].to(DEV, dtype), ch["img"].to(DEV, dtype), ch["uimg"].to(DEV, dtype))
        pos_cat, neg_cat = torch.cat([parts[0], parts[2]], dim=1), torch.cat([parts[1], parts[3]], dim=1)
        with torch.cuda.use_mem_pool(pool):
            ip_embeds = torch.cat([neg_cat, pos_cat], dim=0)
        assert torch.equal(ip_embeds, prepare_ip_embeds(*parts)) and ip_embeds._version == 0
        ptrs.append(ip_embeds.data_ptr())
        latents = ch["lat"].to(DEV, dtype)
        with torch.no_grad():
            for t in scheduler.timesteps:
                latent_model_input = torch.cat([latents] * 2)
                latent_model_input = scheduler.scale_model_input(latent_model_input, t)
                noise_pred = unet(latent_model_input.half() if dtype == torch.float16 else latent_model_input, t,
                                  encoder_hidden_states=ip_embeds, cross_attention_kwargs=None, return_dict=False)[0]
                noise_pred_uncond, noise_pred_text = noise_pred.chunk(2)
                noise_pred = noise_pred_uncond + gs * (noise_pred_text - noise_pred_uncond)
                latents = scheduler.step(noise_pred, t, latents).prev_sample
        results.append(latents.float().cpu())
        del ip_embeds, latent_model_input, noise_pred, noise_pred_uncond, noise_pred_text
    # the hazard this test exists for: the second character's embeddings live where the first one's did
    assert ptrs[0] == ptrs[1], "allocator did not reuse the freed block: the aliasing case was not exercised"
    for c, ch in enumerate(chars):
        enc = torch.cat([torch.cat([ch["neg"], ch["uimg"]], 1), torch.cat([ch["pos"], ch["img"]], 1)], 0).to(dtype).float()
        ref = ch["lat"].to(dtype).float()
        for t in osch.timesteps.tolist():
            npred = ou.unet_forward(cfg, sd_r, torch.cat([ref] * 2), t, enc, ip_scale=0.4).to(dtype).float()
            u, cnd = npred.chunk(2)
            ref = osch.step((u + gs * (cnd - u)).to(dtype).float(), t, ref).to(dtype).float()
        # fp16 storage everywhere (latents, noise_pred), 4 CFG steps of the tiny plan
        pm.check(results[c], ref, f"reference-shaped loop, character {c} (fresh embeddings at a reused address)", 8.0e-3, 2.5e-2)
    # and the two characters really are different conditionings (a stale K/V hit would make B follow A's prompts)
    assert pm.metrics(results[1], results[0])["rel_l2"] > 0.05

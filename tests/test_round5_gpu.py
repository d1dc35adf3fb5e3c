"""GPU: round-5 parity additions (VERDICT r4 "next" item 3).

  * the ROW-CHAIN kernels of round 4 (``tg_rc_xattn``, ``tg_rc_linear``, ``tg_rc_ff``, ``tg_rc_front``) against fixtures captured from the
    IMPORTED reference (tests/golden/block.npz, ``make_golden.py::gen_block``: SD-1.5 first-level geometry — 320 channels = 8 heads x 40,
    77 + {4, 16} tokens, IP scales 0 / 0.1 / 0.4 / 1.0 — composed as models/attention.py:186-236 and models/transformer_2d.py:285-327 compose the
    reference's Attention / IPAttnProcessor / FeedForward).  ``rowchain.MIN_ROWS`` / ``MIN_ROWS_CHAIN`` are lowered so that the 256-row
    fixtures take the fused launches, and every test ASSERTS that the fused launch ran (a counting wrapper around the op);
  * one gated-scale loop step of ``custom_pipelines.StableDiffusionXLCustomPipeline`` on the FULL SDXL plan vs the oracle loop;
  * ``IPAttnProcessor.scale`` assigned while engines replay on their own streams (two engines, one UNet): the device scalar is written on the
    stream the replay is ordered after.
"""
import os

import numpy as np
import pytest
import torch

from tests.golden import gen_common as gc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DTYPES = [torch.bfloat16, torch.float16]


def op_tol(dtype):
    return 1.5e-2 if dtype == torch.bfloat16 else 4e-3


def close(got, ref, tol, what, l2=None):
    from tests import parity_metrics as pm
    got = torch.as_tensor(got).detach().float().cpu()
    ref = torch.as_tensor(ref).detach().float().cpu()
    assert got.shape == ref.shape, f"{what}: {got.shape} vs {ref.shape}"
    return pm.check(got, ref, what, tol / 2 if l2 is None else l2, tol)["max_rel"]


def _load(name):
    return np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))


def _count(monkeypatch, names):
    """wrap ``theatergen_amd.ops.<name>`` so that a test can assert which fused launches ran"""
    from theatergen_amd import ops
    calls = {n: 0 for n in names}
    for n in names:
        orig = getattr(ops, n)

        def wrapper(*a, _o=orig, _n=n, **k):
            calls[_n] += 1
            return _o(*a, **k)
        monkeypatch.setattr(ops, n, wrapper)
    return calls


def _transformer(T, dtype, scale, ip=True):
    from theatergen_amd.attention_processor import IPAttnProcessor
    from theatergen_amd.unet import Transformer2DModel
    sd, x, enc = gc.block_params(T)
    tf = Transformer2DModel(gc.BLOCK_HEADS, gc.BLOCK_C // gc.BLOCK_HEADS, gc.BLOCK_C, 1, gc.BLOCK_CTX, 32, False)
    if ip:
        tf.transformer_blocks[0].attn2.set_processor(IPAttnProcessor(gc.BLOCK_C, gc.BLOCK_CTX, scale=scale, num_tokens=T))
    else:
        sd = {k: v for k, v in sd.items() if ".processor." not in k}
    tf.load_state_dict(sd, strict=True)
    return tf.to(DEV, dtype), x, enc


def _lower_thresholds(monkeypatch):
    from theatergen_amd import rowchain, unet
    monkeypatch.setattr(unet, "_FUSE_LN_MIN_ROWS", 128)       # the block's LayerNorm-folded dispatch (unet.py: BasicTransformerBlock.run)
    monkeypatch.setattr(rowchain, "ENABLED", True)
    monkeypatch.setattr(rowchain, "MODE", 15)
    monkeypatch.setattr(rowchain, "MIN_ROWS", 128)
    monkeypatch.setattr(rowchain, "MIN_ROWS_CHAIN", 128)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T", gc.BLOCK_T)
def test_rc_xattn_vs_reference_golden(dtype, T, monkeypatch):
    """norm2 + IPAttnProcessor(attn2) + residual of a first-level block THROUGH ``fused_cross_block`` / ``tg_rc_xattn`` vs the imported reference,
    all four IP scales (the scale is a device scalar: one packed weight set, four launches), and the plain-AttnProcessor (text-only) instance"""
    from theatergen_amd.attention_processor import AttnProcessor, fused_cross_block
    gold = _load("block")
    _lower_thresholds(monkeypatch)
    calls = _count(monkeypatch, ["rc_xattn", "attention"])
    tf, x, enc = _transformer(T, dtype, 0.4)
    blk = tf.transformer_blocks[0]
    B, C, hh, ww = x.shape
    tok = x.permute(0, 2, 3, 1).reshape(B * hh * ww, C).contiguous().to(DEV, dtype)
    encd = enc.to(DEV, dtype)
    for s in gc.IP_SCALES:
        blk.attn2.processor.scale = s
        got = fused_cross_block(blk.attn2, blk.norm2, tok, B, hh * ww, encd, {})
        assert got is not None, "the fixture must take the fused launch"
        close(got.reshape(B, hh * ww, C), gold[f"T{T}.xattn.scale{s}"], op_tol(dtype), f"rc_xattn T{T} scale {s} {dtype}")
    assert calls["rc_xattn"] == len(gc.IP_SCALES) and calls["attention"] == 0
    blk.attn2.set_processor(AttnProcessor())
    got = fused_cross_block(blk.attn2, blk.norm2, tok, B, hh * ww, encd[:, :77].contiguous(), {})
    assert got is not None and calls["rc_xattn"] == len(gc.IP_SCALES) + 1
    close(got.reshape(B, hh * ww, C), gold[f"T{T}.xattn.plain"], op_tol(dtype), f"rc_xattn T{T} plain {dtype}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T", gc.BLOCK_T)
def test_first_level_block_and_transformer_row_chains_vs_reference_golden(dtype, T, monkeypatch):
    """the whole BasicTransformerBlock (rc_linear to_out, rc_xattn, rc_ff) and the whole Transformer2DModel (rc_front, ..., rc_ff with proj_out +
    residual) on the row-chain launches vs the imported reference's composition; then the same fixtures with the row chains OFF (the path the
    older goldens pin) — both sides of the switch against ONE reference value"""
    from theatergen_amd import rowchain
    from theatergen_amd.unet import _Act
    gold = _load("block")
    _lower_thresholds(monkeypatch)
    calls = _count(monkeypatch, ["rc_xattn", "rc_linear", "rc_ff", "rc_front"])
    tf, x, enc = _transformer(T, dtype, 0.4)
    blk = tf.transformer_blocks[0]
    B, C, hh, ww = x.shape
    n = hh * ww
    tok = x.permute(0, 2, 3, 1).reshape(B * n, C).contiguous().to(DEV, dtype)
    encd = enc.to(DEV, dtype)
    tol = op_tol(dtype)
    got = blk.run(tok, B, n, encd, {})
    assert calls["rc_xattn"] == 1 and calls["rc_ff"] == 1 and calls["rc_linear"] == 1 and calls["rc_front"] == 0, calls
    close(got.reshape(B, n, C), gold[f"T{T}.block.scale0.4"], 1.5 * tol, f"block row-chain T{T} {dtype}")
    out = tf.run(_Act(tok, B, hh, ww, C), encd, {}).t
    assert calls["rc_front"] == 1 and calls["rc_ff"] == 2 and calls["rc_xattn"] == 2, calls
    ref = torch.from_numpy(gold[f"T{T}.transformer.scale0.4"]).permute(0, 2, 3, 1).reshape(B * n, C)
    close(out, ref, 2 * tol, f"transformer row-chain T{T} {dtype}")
    monkeypatch.setattr(rowchain, "ENABLED", False)
    before = dict(calls)
    got = blk.run(tok, B, n, encd, {})
    close(got.reshape(B, n, C), gold[f"T{T}.block.scale0.4"], 1.5 * tol, f"block tiled path T{T} {dtype}")
    out = tf.run(_Act(tok, B, hh, ww, C), encd, {}).t
    close(out, ref, 2 * tol, f"transformer tiled path T{T} {dtype}")
    assert calls == before


def test_sdxl_custom_pipeline_gated_step_on_the_full_plan_vs_oracle():
    """ip_adapter/custom_pipelines.py:308-367 on the FULL SDXL-base plan (1024 x 1024 -> 128 x 128 latents, 2.6 B
    parameters, text_time conditioning, IP-Adapter-Plus = 16 image tokens, fp16): a two-step loop whose first step runs with the IP scale gated
    to 0 (``control_guidance_start`` = 0.5) and whose second step runs with the conditioning scale — the same captured step graph replayed
    with the device scalar changed in between — against the oracle loop."""
    import gc as _gc
    from oracle import ddim as oddim
    from oracle import unet as ou
    from tests import parity_metrics as pm
    from tests.test_hotpath_gpu import _build
    from theatergen_amd import config
    from theatergen_amd.custom_pipelines import StableDiffusionXLCustomPipeline
    dtype, T, steps, gs, scale = torch.float16, 16, 2, 5.0, 0.6
    cfg = config.PLANS["sdxl"]()
    unet, sd_r = _build(cfg, dtype, T=T, scale=scale)
    g = torch.Generator().manual_seed(71)
    ctx = cfg.cross_attention_dim
    pos, neg = torch.randn(1, 77 + T, ctx, generator=g) * 0.5, torch.randn(1, 77 + T, ctx, generator=g) * 0.5
    pooled, npooled = torch.randn(1, 1280, generator=g), torch.randn(1, 1280, generator=g)
    lat = torch.randn(1, 4, 128, 128, generator=g)
    pipe = StableDiffusionXLCustomPipeline(unet)
    pipe.set_scale(scale)
    seen = []
    out = pipe(prompt_embeds=pos.to(DEV, dtype), negative_prompt_embeds=neg.to(DEV, dtype), pooled_prompt_embeds=pooled.to(DEV, dtype),
               negative_pooled_prompt_embeds=npooled.to(DEV, dtype), height=1024, width=1024, num_inference_steps=steps, guidance_scale=gs,
               latents=lat, control_guidance_start=0.5, control_guidance_end=1.0,
               callback=lambda i, t, x: seen.append((i, t, x.detach().float().cpu().clone()))).images.float().cpu()
    assert [i for i, _, _ in seen] == [0, 1]
    del pipe, unet
    _gc.collect()
    torch.cuda.empty_cache()
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    osch = oddim.DDIMSchedule()
    osch.set_timesteps(steps)
    assert [t for _, t, _ in seen] == osch.timesteps.tolist()
    enc = torch.cat([neg, pos]).to(dtype).float()
    added = {"text_embeds": torch.cat([npooled, pooled]).to(dtype).float(), "time_ids": torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]] * 2)}
    ref = lat.clone()
    for i, t in enumerate(osch.timesteps.tolist()):
        s = 0.0 if i / steps < 0.5 else scale
        npred = ou.unet_forward(cfg, sd_r, torch.cat([ref] * 2).to(dtype).float(), t, enc, ip_scale=s, num_tokens=T, added_cond_kwargs=added)
        ref = oddim.step_epilogue(osch, npred, t, ref, gs)
        pm.check(seen[i][2].reshape(ref.shape), ref, f"SDXL full plan, custom-pipeline step {i} (IP scale {s}), fp16", 7.5e-3, 1.5e-2)
    pm.check(out.reshape(ref.shape), ref, "SDXL full plan, custom-pipeline loop output, fp16", 7.5e-3, 1.5e-2)


def test_ip_scale_assignment_is_fenced_against_engine_streams():
    """``IPAttnProcessor.scale = s`` while two engines replay on THEIR OWN streams.  The device scalar the
    attention kernels read is filled on the current stream; the fill now waits for the replays already queued on the registered engine streams and
    the next replays wait for it, so gating the scale between two rounds of ``run_concurrent`` gives the bits of the same gating on a single stream."""
    from tests.test_hotpath_gpu import _build
    from theatergen_amd import config
    from theatergen_amd.attention_processor import IPAttnProcessor
    from theatergen_amd.pipelines import DenoiseEngine
    dtype = torch.bfloat16
    cfg = config.tiny()
    unet, _ = _build(cfg, dtype)
    procs = [p for p in unet.attn_processors.values() if isinstance(p, IPAttnProcessor)]
    assert procs
    g = torch.Generator().manual_seed(41)
    steps = 6
    lats = [torch.randn(2, 4, 16, 16, generator=g) for _ in range(2)]
    encs = [torch.randn(4, 81, cfg.cross_attention_dim, generator=g) * 0.5 for _ in range(2)]
    engs = [DenoiseEngine(unet, None, n_img=2, height=128, width=128, num_inference_steps=steps, guidance_scale=7.5, enc_len=81) for _ in range(2)]
    for e, enc in zip(engs, encs):
        e.set_conditioning(enc.to(DEV, dtype))

    def gate(i):                                                  # ip_adapter/custom_pipelines.py:328-333 with start 1/3, end 2/3
        s = 0.0 if (i / steps < 1 / 3 or (i + 1) / steps > 2 / 3) else 0.7
        for p in procs:
            p.scale = s
    seq = [e.run(lat, before_step=gate).clone() for e, lat in zip(engs, lats)]
    ungated = []
    for p in procs:
        p.scale = 0.7
    ungated = [e.run(lat).clone() for e, lat in zip(engs, lats)]
    assert not torch.equal(seq[0][-1], ungated[0][-1]), "the gating must change the result for the test to mean anything"
    for rep in range(3):
        par = [h.clone() for h in DenoiseEngine.run_concurrent(engs, lats, before_step=gate)]
        torch.cuda.synchronize()
        for k in range(2):
            assert torch.equal(seq[k], par[k]), f"gated concurrent replay differs from the gated single-stream run (engine {k}, repeat {rep})"


def test_row_chain_dev_switches_are_rejected_in_release_calls():
    """the timing switches of the row-chain kernels (skip stores / barriers / MFMAs) ride in descriptor bits; without TG_RC_DEV=1 in the
    environment a descriptor that carries them is an argument error, not a silently wrong result"""
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import rc_pack
    if os.environ.get("TG_RC_DEV") == "1":
        pytest.skip("dev switches enabled in this environment")
    x = torch.randn(256, 320, device=DEV).to(torch.bfloat16)
    w = (torch.randn(320, 320, device=DEV) / 18).to(torch.bfloat16)
    wpk = rc_pack(w, None)
    ops.rc_linear(x, wpk, 320)
    with pytest.raises(RuntimeError):
        ops.rc_linear(x, wpk, 320, variant=256)


def test_skinny_gemm_large_k_instances_after_the_register_fix():
    """round 5: the K / 128 in {16, 20} (LayerNorm-folded) and K = 5120 launches of tg_skinny_gemm were the instances that spilled; their chunk
    structure changed (weight fragments in two pieces, fp32 row conversions not kept alive): parity vs the fp32 reference of the op"""
    import torch.nn.functional as F
    from tests import parity_metrics as pm
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import pack_ln_linear, skinny_pack
    for dtype, (l2, mx) in ((torch.bfloat16, (3e-3, 1e-2)), (torch.float16, (4e-4, 2.5e-3))):
        for K, N, ln in ((2048, 256, True), (2560, 128, True), (5120, 1280, False), (2048, 2048, False), (1536, 64, True)):
            g = torch.Generator().manual_seed(K + N)
            x = (torch.randn(32, K, generator=g) * 1.3 + 0.2).to(dtype).to(DEV)
            W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype).to(DEV)
            b = torch.randn(N, generator=g).to(dtype).to(DEV)
            if ln:
                gamma = (1 + 0.2 * torch.randn(K, generator=g)).to(dtype).to(DEV)
                beta = (0.1 * torch.randn(K, generator=g)).to(dtype).to(DEV)
                Wp, u, v = pack_ln_linear(W, b, gamma, beta)
                got = ops.skinny_gemm(x, skinny_pack(Wp), N, ln=(u, v, 1e-5))
                ref = F.linear(F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-5), W.float(), b.float())
            else:
                got = ops.skinny_gemm(x, skinny_pack(W), N, bias=b)
                ref = F.linear(x.float(), W.float(), b.float())
            pm.check(got.float().cpu(), ref.cpu(), f"skinny K{K} N{N} ln={ln} {dtype}", l2 * (1.5 if ln else 1), mx * (1.5 if ln else 1))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(16384, 640, 640), (4096, 1280, 1280), (9000, 1000, 192), (300, 640, 64), (4096, 1280, 5120), (66000, 320, 320)])
def test_gemm_128x160_tiles(dtype, shape):
    """round 5: gemm_glds_kernel on 128 x 160 tiles (force_tile 21 / 23 = BK 64 x 3 stages / BK 64 x 2 stages; tg_gemm_t160.hip) —
    ragged M and N, K of one tile up to 80 tiles, bias + per-batch vector + residual through the chunked LDS epilogue; the K order per output is the
    128 x 128 kernel's, so the results must be BIT-identical to it; and the planner's own choice (force_tile 0) for the shapes it was built for."""
    import math
    from tests.test_kernels_gpu import check, rnd
    from theatergen_amd import ops
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    rows = 100 if M % 100 == 0 else M
    a, w = rnd((M, K), dtype, g), rnd((N, K), dtype, g, 1 / math.sqrt(K))
    bias, res, bvec = rnd((N,), dtype, g), rnd((M, N), dtype, g), rnd((M // rows, N), dtype, g)
    ad, wd, bd, rd, vd = a.to(DEV), w.to(DEV), bias.to(DEV), res.to(DEV), bvec.to(DEV)
    ref = (ad.float() @ wd.float().t() + bd.float() + rd.float() + vd.float().repeat_interleave(rows, 0)).cpu()
    base = ops.linear(ad, wd, bd, res=rd, bvec=vd, rows_per_batch=rows, force_tile=1)
    # round 6: the planner hands long-K whole-round problems (here 16384 x 640 x 640) to the ping-pong 256 x 160 kernel (kernel_kind 7: 16 x 16 x 32 MFMAs,
    # another K order) — its own tests are in test_round6_gpu.py; bit-identity with the 128 x 128 kernel is a property of the round-5 tiles only
    kind0 = ops.gemm(ad, wd, M, N, K, bias=bd, res=rd, bvec=vd, rows_per_batch=rows, plan_only=True)[3]
    for tile in (21, 23, 0):
        out = ops.linear(ad, wd, bd, res=rd, bvec=vd, rows_per_batch=rows, force_tile=tile)
        check(out, ref, dtype, f"128x160 tile {tile} {shape}")
        if tile != 0 or kind0 != 7:
            assert torch.equal(out, base), f"tile {tile} {shape}: not bit-identical to the 128 x 128 kernel"
    if (M, N) in ((16384, 640), (4096, 1280)):
        pl = ops.gemm(ad, wd, M, N, K, bias=bd, res=rd, plan_only=True)
        assert (pl[0], pl[1]) == ((256, 160) if pl[3] == 7 else (128, 160)), pl


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,C", [(16384, 640), (4096, 1280), (65536, 320)])
def test_gemm_layernorm_folded_on_128x160_tiles(dtype, M, C, monkeypatch):
    """round 5: the LayerNorm-folded projections (attn2.to_q, attn1 q | k | v^T) on the tile-count-aware 128 x 160 tiles vs
    ``F.linear(F.layer_norm(x), W)`` in fp32, in-kernel and precomputed row statistics, and BIT-identical to the 128 x 128 instance (TG_T160=0):
    the fold runs in accumulator layout with the same operands in the same order."""
    import math
    import torch.nn.functional as F
    from tests.test_kernels_gpu import check, rnd
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import pack_ln_linear
    g = torch.Generator().manual_seed(M + C)
    x = ((torch.randn(M, C, generator=g) + 0.7 * torch.randn(M, 1, generator=g)) * 1.3).to(dtype)
    gamma, beta = (1 + 0.3 * torch.randn(C, generator=g)).to(dtype), (0.3 * torch.randn(C, generator=g)).to(dtype)
    eps = 1e-5
    xn = F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), eps)
    xd = x.to(DEV)
    w = rnd((C, C), dtype, g, 1 / math.sqrt(C))
    wl, u, v = pack_ln_linear(w.to(DEV), None, gamma.to(DEV), beta.to(DEV))
    on160 = ops.gemm(xd, wl, M, C, C, ln=(u, v, eps), plan_only=True)[:2] == (128, 160)
    assert on160 or C == 320                                 # (65536 x 320 fills whole rounds with 128 x 128 already: planner keeps it)
    got = ops.gemm(xd, wl, M, C, C, ln=(u, v, eps))
    check(got, xn @ w.float().t(), dtype, f"ln-folded to_q 128x160 {(M, C)}", scale=1.5)
    rows_st = ops.layernorm_stats(xd, eps)
    got2 = ops.gemm(xd, wl, M, C, C, ln=(u, v, eps, rows_st))
    check(got2, xn @ w.float().t(), dtype, f"ln-folded (precomputed statistics) to_q 128x160 {(M, C)}", scale=1.5)
    B = 2
    rows = M // B
    w3 = rnd((3 * C, C), dtype, g, 1 / math.sqrt(C))
    wl3, u3, v3 = pack_ln_linear(w3.to(DEV), None, gamma.to(DEV), beta.to(DEV))

    def qkv():
        qk = torch.zeros((M, 2 * C), dtype=dtype, device=DEV)
        vt = torch.zeros((B, C, rows), dtype=dtype, device=DEV)
        ops.gemm(xd, wl3, M, 3 * C, C, rows_per_batch=rows, out=qk, n_split=2 * C, out_t=vt, ldt=rows, ln=(u3, v3, eps))
        return qk, vt
    assert ops.gemm(xd, wl3, M, 3 * C, C, rows_per_batch=rows, out=torch.empty((M, 2 * C), dtype=dtype, device=DEV), n_split=2 * C,
                    out_t=torch.empty((B, C, rows), dtype=dtype, device=DEV), ldt=rows, ln=(u3, v3, eps), plan_only=True)[:2] == (128, 128)   # planner keeps q | k | v^T on 128 x 128
    qk, vt = qkv()
    ref3 = xn @ w3.float().t()
    check(qk, ref3[:, :2 * C], dtype, f"ln-folded q|k 128x160 {(M, C)}", scale=1.5)
    check(vt, ref3[:, 2 * C:].reshape(B, rows, C).permute(0, 2, 1), dtype, f"ln-folded v^T 128x160 {(M, C)}", scale=1.5)
    monkeypatch.setenv("TG_T160", "0")
    assert ops.gemm(xd, wl, M, C, C, ln=(u, v, eps), plan_only=True)[:2] == (128, 128)
    # with PRECOMPUTED statistics both instances fold the same fp32 (rstd, -rstd mean) into the same MFMA sums: bit-identical.  The in-kernel statistics are
    # fp32 sums over the K-tiles a workgroup stages (64-wide here, 32-wide in the K <= 640 instance of the 128 x 128 kernel): same value to fp32 rounding
    old2 = ops.gemm(xd, wl, M, C, C, ln=(u, v, eps, rows_st))
    assert torch.equal(old2, got2), "128 x 160 and 128 x 128 LayerNorm-folded instances differ with identical row statistics"
    old = ops.gemm(xd, wl, M, C, C, ln=(u, v, eps))
    qk0, vt0 = qkv()
    check(old, got.float(), dtype, f"ln-folded 128x160 vs 128x128 {(M, C)}", scale=0.5)
    check(qk0, qk.float(), dtype, f"ln-folded q|k 128x160 vs 128x128 {(M, C)}", scale=0.5)
    check(vt0, vt.float(), dtype, f"ln-folded v^T 128x160 vs 128x128 {(M, C)}", scale=0.5)


def _xq_reference(x, gamma, beta, wq, k, v, kip, vip, heads, scale, ip_w):
    """fp32 restatement of norm2 -> to_q -> (decoupled) cross-attention of ip_adapter/attention_processor.py:445-529 (O before to_out)"""
    import torch.nn.functional as F
    M, C = x.shape
    B = k.shape[0]
    N = M // B
    d = C // heads
    q = (F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), 1e-5) @ wq.float().t()).reshape(B, N, heads, d).permute(0, 2, 1, 3)
    hd = lambda t: t.float().reshape(B, t.shape[1], heads, d).permute(0, 2, 1, 3)
    o = torch.softmax(q @ hd(k).transpose(-1, -2) * scale, -1) @ hd(v)
    if kip is not None:
        o = o + ip_w * torch.softmax(q @ hd(kip).transpose(-1, -2) * scale, -1) @ hd(vip)
    return o.permute(0, 2, 1, 3).reshape(M, C)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,B,N,T,ip_w", [(640, 2, 256, 4, 0.4), (640, 16, 1024, 4, 0.4), (640, 3, 128, 0, 0.0), (640, 2, 128, 16, 1.0),
                                          (1280, 2, 256, 4, 0.4), (1280, 16, 256, 16, 0.7), (1280, 2, 128, 0, 0.0), (320, 2, 128, 4, 0.4)])
def test_xq_attn_vs_fp32_reference(dtype, C, B, N, T, ip_w):
    """tg_xq_attn + tg_xq_kv_pack: norm2 + to_q + the two softmaxes + PV in one launch (128 x 160 tiles, two heads of 80 / one head of 160 per tile; both
    K-stage instances: <= 256 tiles and more) vs the fp32 restatement; the IP scale read from the device at run time; C = 320 is refused (head dim 40)."""
    import math
    from tests import parity_metrics as pm
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import pack_ln_linear
    heads, L = 8, 77
    d = C // heads
    M = B * N
    g = torch.Generator().manual_seed(C + B + N + T)
    t = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    x = (t(M, C, sc=1.2) + 0.2).to(dtype).to(DEV)
    wq = t(C, C, sc=C ** -0.5).to(dtype).to(DEV)
    gamma, beta = (1 + 0.2 * t(C)).to(dtype).to(DEV), t(C, sc=0.1).to(dtype).to(DEV)
    k, v = t(B, L, C).to(dtype).to(DEV), t(B, L, C).to(dtype).to(DEV)
    kip, vip = (t(B, T, C).to(dtype).to(DEV), t(B, T, C).to(dtype).to(DEV)) if T else (None, None)
    ldt, ldi = 80, 8 * ((max(T, 1) + 7) // 8)
    vt = torch.zeros(B, C, ldt, device=DEV, dtype=dtype); vt[:, :, :L] = v.transpose(1, 2)
    vtip = None
    if T:
        vtip = torch.zeros(B, C, ldi, device=DEV, dtype=dtype); vtip[:, :, :T] = vip.transpose(1, 2)
    scale = d ** -0.5
    wl, u, vv = pack_ln_linear(wq, None, gamma, beta, scale=scale * math.log2(math.e))
    if d not in (80, 160):
        with pytest.raises(RuntimeError):
            ops.xq_kv_pack(k.reshape(B * L, C), vt, ldt, L, None, None, 0, 0, B, C, d)
        return
    blob = ops.xq_kv_pack(k.reshape(B * L, C), vt, ldt, L, kip.reshape(B * T, C) if T else None, vtip, ldi, T, B, C, d)
    w = torch.full((1,), ip_w, device=DEV)
    got = ops.xq_attn(x, wl, u, vv, 1e-5, blob, d, N, L, T, ip_scale=w if T else None)
    ref = _xq_reference(x, gamma, beta, wq, k, v, kip, vip, heads, scale, ip_w)
    l2, mx = (4e-3, 1.2e-2) if dtype == torch.bfloat16 else (6e-4, 3e-3)       # q, P and O are storage-dtype roundings on the way (rc_xattn's bar x 1.2)
    pm.check(got.float().cpu(), ref.cpu(), f"xq_attn C{C} B{B} N{N} T{T} w{ip_w} {dtype}", l2, mx)
    assert torch.equal(got, ops.xq_attn(x, wl, u, vv, 1e-5, blob, d, N, L, T, ip_scale=w if T else None))
    if T:
        w.fill_(0.0)                                       # the same launch with another device-side scale (graph-replay contract)
        got0 = ops.xq_attn(x, wl, u, vv, 1e-5, blob, d, N, L, T, ip_scale=w)
        pm.check(got0.float().cpu(), _xq_reference(x, gamma, beta, wq, k, v, kip, vip, heads, scale, 0.0).cpu(), f"xq_attn scale 0 C{C} {dtype}", l2, mx)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ci", range(len(gc.XQ_CASES)))
def test_inner_level_cross_attention_fused_vs_reference_golden(dtype, ci, monkeypatch):
    """the inner-level cross-attention sub-block (C = 640 / 8 x 80 and C = 1280 / 8 x 160) THROUGH the fused launch — our processors called
    the way BasicTransformerBlock calls them (LayerNorm handed over for folding) — vs the imported reference's Attention + IPAttnProcessor / AttnProcessor on
    torch's LayerNorm output (tests/golden/xq.npz); then the same call with the fused launch off (three launches): both against ONE reference value."""
    from theatergen_amd import attention_processor as AP
    gold = _load("xq")
    name, C, heads, ctx, N, T, scale, ip = gc.XQ_CASES[ci]
    w, norm_sd, x, enc = gc.xq_params(ci)
    attn = AP.Attention(query_dim=C, cross_attention_dim=ctx, heads=heads, dim_head=C // heads)
    attn.load_state_dict({k: v for k, v in w.items() if "_ip" not in k})
    if ip:
        proc = AP.IPAttnProcessor(hidden_size=C, cross_attention_dim=ctx, scale=scale, num_tokens=T)
        proc.load_state_dict({"to_k_ip.weight": w["to_k_ip.weight"], "to_v_ip.weight": w["to_v_ip.weight"]})
        attn.set_processor(proc)
    attn = attn.to(DEV, dtype)
    norm = torch.nn.LayerNorm(C)
    norm.load_state_dict(norm_sd)
    norm = norm.to(DEV, dtype)
    xd, encd = x.to(DEV, dtype), enc.to(DEV, dtype)
    monkeypatch.setattr(AP, "XQ_MIN_ROWS", 128)
    calls = _count(monkeypatch, ["xq_attn", "attention"])
    got = attn.processor(attn, xd, encoder_hidden_states=encd, _fused_ln=(norm, None))
    assert calls["xq_attn"] == 1 and calls["attention"] == 0, calls
    close(got, gold[f"{name}.out"], op_tol(dtype), f"fused inner cross-attention {name} {dtype}")
    monkeypatch.setattr(AP, "XQ_ENABLED", False)
    old = attn.processor(attn, xd, encoder_hidden_states=encd, _fused_ln=(norm, None))
    assert calls["xq_attn"] == 1 and calls["attention"] == 1, calls
    close(old, gold[f"{name}.out"], op_tol(dtype), f"three-launch inner cross-attention {name} {dtype}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(4096, 1280, 5120), (16384, 640, 2560), (1000, 320, 1280), (256, 128, 64)])
def test_gemm_padded_row_pitches(dtype, M, N, K):
    """round 5: ``lda`` / ``ldw`` of tg_gemm (plain single-source GEMM): A and W as [:, :K] views of buffers whose rows are 64 elements longer (what
    unet.FeedForward does with its hidden tensor and net.2's weight) — same bits as the contiguous operands, every tile the planner may pick"""
    import math
    from tests.test_kernels_gpu import check, rnd
    from theatergen_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a, w = rnd((M, K), dtype, g).to(DEV), rnd((N, K), dtype, g, 1 / math.sqrt(K)).to(DEV)
    bias, res = rnd((N,), dtype, g).to(DEV), rnd((M, N), dtype, g).to(DEV)
    ap = torch.full((M, K + 64), float("nan"), dtype=dtype, device=DEV); ap[:, :K] = a
    wp = torch.full((N, K + 64), float("nan"), dtype=dtype, device=DEV); wp[:, :K] = w
    ref = (a.float() @ w.float().t() + bias.float() + res.float()).cpu()
    for tile in (0, 1, 7, 21, 23):
        base = ops.linear(a, w, bias, res=res, force_tile=tile)
        got = ops.linear(ap[:, :K], wp[:, :K], bias, res=res, force_tile=tile)
        check(got, ref, dtype, f"padded pitches {(M, N, K)} tile {tile}")
        assert torch.equal(got, base), f"tile {tile}: padded operands differ from contiguous ones"
    with pytest.raises(RuntimeError):
        ops.gemm(ap[:, :K], wp[:, :K], M, N, K, lda=K + 64, ldw=K + 64, force_tile=10)          # the big tile does not take padded pitches

@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,h,w,cout,sdt", [(16, 64, 64, 320, torch.float32), (2, 96, 96, 320, torch.float32), (3, 17, 23, 320, torch.float32),
                                            (2, 128, 128, 320, None), (1, 8, 8, 160, torch.float32), (2, 16, 16, 640, torch.float32)])
def test_conv_in_on_matrix_cores(dtype, B, h, w, cout, sdt):
    """round 5: conv_in (4 -> 320, UNet2DConditionModel.conv_in; models/unet_2d_condition.py:820) as three MFMA k-steps per 32-pixel block with the fp32
    sample split into a storage-dtype head + tail — vs fp32 F.conv2d on the UNROUNDED sample: the result must be the correctly rounded fp32 convolution
    up to fp32 summation order (<= 1 ulp of the storage dtype, almost everywhere exact), ragged pixel counts, image borders, bias."""
    import torch.nn.functional as F
    from tests.test_kernels_gpu import rnd
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import pack_conv3x3
    g = torch.Generator().manual_seed(B + h + w + cout)
    sdt = sdt or dtype
    smp = (torch.randn(B, 4, h, w, generator=g) * 3.0).to(sdt)
    wt, bias = rnd((cout, 4, 3, 3), dtype, g, 1 / 6), rnd((cout,), dtype, g)
    ref = F.conv2d(smp.double(), wt.double(), bias.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
    out = ops.conv_in(smp.to(DEV), pack_conv3x3(wt).to(DEV), bias.to(DEV), cout, dtype).cpu()
    assert out.shape == ref.shape
    want = ref.float().to(dtype)
    ulp = 2.0 ** (-7 if dtype == torch.bfloat16 else -10)
    err = (out.double() - ref).abs()
    # (absolute term: the sample enters as head + tail = 16 (bf16) / 22 (f16) mantissa bits, 36 products of magnitude ~0.5 per output)
    bad = err > ulp * ref.abs() * 1.01 + 5e-4
    assert not bool(bad.any()), (err[bad].max().item(), ref[bad].abs().min().item())
    assert (out == want).float().mean().item() > 0.95


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,h,w,cin,cout,odt", [(16, 64, 64, 320, 4, torch.float32), (2, 96, 96, 320, 4, torch.float32), (2, 128, 128, 320, 4, None),
                                                (3, 8, 16, 64, 4, torch.float32), (2, 16, 32, 128, 8, torch.float32), (1, 24, 48, 256, 3, None)])
def test_conv_out_on_matrix_cores_with_groupnorm_prologue(dtype, B, h, w, cin, cout, odt):
    """round 5: conv_norm_out + SiLU + conv_out (models/unet_2d_condition.py:1015-1018) in one launch on the matrix cores: vs fp32
    conv2d(silu(group_norm(x))) ; BIT-identical to tg_groupnorm followed by the plain tg_conv_out (same expression, same rounding of the normalised
    activations, same kernel); image borders, tiles of several images, cout up to 8, both output dtypes."""
    import torch.nn.functional as F
    from tests.test_kernels_gpu import rnd
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import pack_conv3x3
    g = torch.Generator().manual_seed(B + h + w + cin + cout)
    odt = odt or dtype
    x = (rnd((B, cin, h, w), dtype, g) * 1.5 + 0.3)
    gamma, beta = (1 + 0.2 * torch.randn(cin, generator=g)).to(dtype), (0.1 * torch.randn(cin, generator=g)).to(dtype)
    wt, bias = rnd((cout, cin, 3, 3), dtype, g, 1 / (3 * cin ** 0.5)), rnd((cout,), dtype, g)
    assert ops.conv_out_takes_gn(cin, h, w, cout)
    tok = x.permute(0, 2, 3, 1).reshape(B * h * w, cin).contiguous().to(DEV)
    wp = pack_conv3x3(wt).to(DEV)
    coef = ops.groupnorm_coef(tok, B, h * w, 32, 1e-5, gamma.to(DEV), beta.to(DEV))
    fused = ops.conv_out(tok, wp, bias.to(DEV), B, h, w, cout, odt, coef=coef, silu=True)
    y = ops.groupnorm(tok, B, h * w, 32, 1e-5, gamma.to(DEV), beta.to(DEV), silu=True)
    two = ops.conv_out(y, wp, bias.to(DEV), B, h, w, cout, odt)
    assert torch.equal(fused, two), "conv_out with the GroupNorm prologue differs from tg_groupnorm + tg_conv_out"
    # fp32 reference on the ROUNDED normalised activations (what both paths feed the conv) and, looser, on the unrounded ones
    yn = y.float().cpu().reshape(B, h, w, cin).permute(0, 3, 1, 2)
    ref = F.conv2d(yn.double(), wt.double(), bias.double(), padding=1)
    err = (fused.double().cpu() - ref).abs().max().item()
    tol = 2e-3 if odt == torch.float32 else (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -9) * max(1.0, ref.abs().max().item())
    assert err <= tol, (err, tol)
    full = F.conv2d(F.silu(F.group_norm(x.float(), 32, gamma.float(), beta.float(), 1e-5)), wt.float(), bias.float(), padding=1)
    assert (fused.float().cpu() - full).abs().max().item() <= 0.05 * max(1.0, full.abs().max().item())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N,heads,d", [(2, 576, 5, 64), (1, 2304, 5, 64), (2, 200, 3, 40), (1, 136, 2, 64), (2, 64, 4, 32), (1, 8, 1, 8), (1, 1096, 10, 64)])
def test_attention_reverse_pass_without_materialised_probabilities(dtype, B, N, heads, d):
    """round 5: tg_attention_bwd (statistics / dQ / dK + dV launches of one recompute kernel) vs fp32 autograd of softmax(s Q K^T) V on the same
    storage-dtype operands: ragged last tiles (N not a multiple of 64 / 128), head dims below 64, one-tile problems."""
    from tests import parity_metrics as pm
    from theatergen_amd import ops
    g = torch.Generator().manual_seed(B + N + heads + d)
    inner = heads * d
    q, k, v, do = [(torch.randn(B * N, inner, generator=g) * s_).to(dtype) for s_ in (1.0, 1.0, 1.0, 0.5)]
    scale = d ** -0.5
    def heads_(t):
        return t.float().reshape(B, N, heads, d).permute(0, 2, 1, 3)
    qr, kr, vr = [heads_(t).clone().requires_grad_(True) for t in (q, k, v)]
    o = torch.softmax(qr @ kr.transpose(-1, -2) * scale, -1) @ vr
    rq, rk, rv = torch.autograd.grad(o, [qr, kr, vr], heads_(do))
    dq, dk, dv = ops.attention_bwd(q.to(DEV), k.to(DEV), v.to(DEV), do.to(DEV), B, N, heads, d, scale)
    l2, mx = (2e-2, 6e-2) if dtype == torch.bfloat16 else (3e-3, 1.5e-2)
    for name, got, ref in (("dQ", dq, rq), ("dK", dk, rk), ("dV", dv, rv)):
        pm.check(got.float().cpu().reshape(B, N, heads, d).permute(0, 2, 1, 3), ref, f"attention_bwd {name} {(B, N, heads, d)} {dtype}", l2, mx)
    again = ops.attention_bwd(q.to(DEV), k.to(DEV), v.to(DEV), do.to(DEV), B, N, heads, d, scale)
    assert all(torch.equal(a, b) for a, b in zip((dq, dk, dv), again)), "not deterministic"


def test_attention_reverse_pass_matches_the_materialised_path(monkeypatch):
    """the recompute-based self-attention reverse pass (default) and the per-(item, head) materialised one (TG_FLASH_BWD=0) through
    backward.attention_input_grad on an SD-2.1-shaped layer; refusal of shapes the kernel does not take."""
    from tests import parity_metrics as pm
    from theatergen_amd import backward, ops
    from theatergen_amd.attention_processor import Attention, AttnProcessor
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    C, heads, B, N = 320, 5, 2, 576
    attn = Attention(query_dim=C, heads=heads, dim_head=C // heads).to(DEV, dtype)
    for p_ in attn.parameters():
        p_.data = (torch.randn(p_.shape, generator=g) * (0.06 if p_.dim() == 2 else 0.02)).to(DEV, dtype)
    h = (torch.randn(B * N, C, generator=g)).to(dtype).to(DEV)
    dout = (torch.randn(B * N, C, generator=g) * 0.5).to(dtype).to(DEV)
    assert backward.FLASH_BWD and ops.attention_bwd_supported(C // heads, N)
    new = backward.attention_input_grad(attn, AttnProcessor(), h, B, N, None, dout, None)
    monkeypatch.setattr(backward, "FLASH_BWD", False)
    old = backward.attention_input_grad(attn, AttnProcessor(), h, B, N, None, dout, None)
    pm.check(new, old.float(), "flash vs materialised self-attention input grad", 1.5e-2, 6e-2)
    assert not ops.attention_bwd_supported(80, 1024) and not ops.attention_bwd_supported(64, 100)
    x = torch.zeros(100, 64, dtype=dtype, device=DEV)
    with pytest.raises(RuntimeError):
        ops.attention_bwd(x, x, x, x, 1, 100, 1, 64, 0.125)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N,L,heads,d,with_extra", [(2, 576, 77, 5, 64, True), (1, 2304, 77, 10, 64, False), (2, 200, 4, 3, 40, False), (1, 1000, 16, 2, 64, True),
                                                     (2, 64, 93, 4, 32, True)])
def test_cross_attention_reverse_pass_segment(dtype, B, N, L, heads, d, with_extra):
    """round 5: tg_attention_bwd_cross — dQ of one softmax segment over a short constant key set (77 text keys / the IP-Adapter's image keys), with the
    guidance loss's d loss / d P joining dO V^T and a segment weight on dS — vs fp32 autograd of  w * softmax(s Q K^T) V  (+ <extra, P>)."""
    from tests import parity_metrics as pm
    from theatergen_amd import ops
    g = torch.Generator().manual_seed(B + N + L + heads + d)
    inner = heads * d
    q, do = (torch.randn(B * N, inner, generator=g)).to(dtype), (torch.randn(B * N, inner, generator=g) * 0.5).to(dtype)
    k, v = (torch.randn(B * L, inner, generator=g)).to(dtype), (torch.randn(B * L, inner, generator=g)).to(dtype)
    extra = (torch.randn(B, heads, N, L, generator=g) * 0.3) if with_extra else None
    scale, wgt = d ** -0.5, 0.4
    def hd(t, n):
        return t.float().reshape(B, n, heads, d).permute(0, 2, 1, 3)
    qr = hd(q, N).clone().requires_grad_(True)
    P = torch.softmax(qr @ hd(k, L).transpose(-1, -2) * scale, -1)
    outs, grads = [wgt * (P @ hd(v, L))], [hd(do, N)]
    if with_extra:
        outs.append(P)
        grads.append(extra)
    ref = torch.autograd.grad(outs, qr, grads)[0]
    dq = ops.attention_bwd_cross(q.to(DEV), do.to(DEV), k.to(DEV), v.to(DEV), B, N, L, heads, d, scale, scale * wgt,
                                 extra=(extra / wgt).to(DEV) if with_extra else None)
    # (the kernel's dS = ds_scale * P o (dP + extra - D): with a segment weight the caller's extra is per unit of weight)
    l2, mx = (2e-2, 6e-2) if dtype == torch.bfloat16 else (3e-3, 1.5e-2)
    pm.check(dq.float().cpu().reshape(B, N, heads, d).permute(0, 2, 1, 3), ref, f"attention_bwd_cross {(B, N, L, heads, d, with_extra)} {dtype}", l2, mx)

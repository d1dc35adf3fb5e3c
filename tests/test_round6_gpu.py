"""GPU, round 6: the ping-pong 256 x 256 GEMM (csrc/tg_gemm_pp.hip, tg_gemm force_tile 24 / planner kind 7) against a plain PyTorch fp32 reference of
the same op, through the C ABI; the drop-in surface closed this round (compose_latents_with_alignment, the reference-signature stage loops)."""
import math

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
    for (M, N, K, geglu, want) in [(16384, 5120, 640, True, 7), (4096, 10240, 1280, True, 7), (4096, 1280, 1280, False, None), (16384, 640, 640, False, None)]:
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

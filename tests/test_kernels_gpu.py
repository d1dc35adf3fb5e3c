"""GPU: every HIP kernel family against a plain PyTorch fp32 CPU reference of the same op, called through
the C ABI (theatergen_amd.ops -> libtheatergen_hip.so).  Inputs are rounded to the storage dtype first so the
comparison isolates the kernel (fp32 accumulate) from input quantisation.

Tolerances (stated per test): outputs are stored in bf16 (8 mantissa bits) / fp16 (11 bits):
  bf16: |err| <= 1.0e-2 * max|ref| + small abs;  fp16: 2.5e-3 * max|ref|.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float16]


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def rel_tol(dtype):
    return 1.0e-2 if dtype == torch.bfloat16 else 2.5e-3


def l2_tol(dtype):
    """relative L2 error of ONE kernel whose output is rounded once to the storage dtype (bf16: 2^-9 per element,
    uniform -> ~1.1e-3 rms; fp16: 2^-12 -> ~1.4e-4) plus fp32 accumulation-order noise"""
    return 3.0e-3 if dtype == torch.bfloat16 else 4.0e-4


def check(got, ref, dtype, what="", scale=1.0):
    from tests import parity_metrics as pm
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    pm.check(got, ref, f"kernel: {what}", l2_tol(dtype) * scale, rel_tol(dtype) * scale, dtype=str(dtype))


def rnd(shape, dtype, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def test_library_loads_and_reports_version():
    from theatergen_amd import _lib
    assert _lib.lib().tg_version() == _lib.ABI_VERSION


@pytest.mark.parametrize("dtype", DTYPES)
def test_mfma_layout_probe(dtype):
    """A[i][k], B[k][j] fragments laid out as the kernels assume: lane l holds row/col l&31, k = 8*(l>>5)+e."""
    from theatergen_amd import _lib
    dev = _dev()
    g = torch.Generator().manual_seed(1)
    A = rnd((32, 16), dtype, g)
    Bm = rnd((16, 32), dtype, g)
    af = torch.empty(64, 8, dtype=dtype)
    bf = torch.empty(64, 8, dtype=dtype)
    for l in range(64):
        for e in range(8):
            af[l, e] = A[l & 31, 8 * (l >> 5) + e]
            bf[l, e] = Bm[8 * (l >> 5) + e, l & 31]
    d = torch.zeros(64, 16, dtype=torch.float32, device=dev)
    afd, bfd = af.to(dev), bf.to(dev)
    _lib.check(_lib.lib().tg_debug_mfma32(0 if dtype == torch.bfloat16 else 1, afd.data_ptr(), bfd.data_ptr(), d.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    ref = A.float() @ Bm.float()
    got = torch.empty(32, 32)
    dc = d.cpu()
    for l in range(64):
        for r in range(16):
            got[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31] = dc[l, r]
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5)


GEMM_SHAPES = [
    # M, N, K
    (256, 320, 320), (8192, 320, 320), (2, 1280, 320), (154, 640, 768), (130, 1280, 2560), (512, 2560, 320),
    (100, 64, 72),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", GEMM_SHAPES)
def test_gemm_plain(dtype, shape):
    from theatergen_amd import ops
    dev = _dev()
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    a, w = rnd((M, K), dtype, g), rnd((N, K), dtype, g, 1 / math.sqrt(K))
    bias = rnd((N,), dtype, g)
    ref = a.float() @ w.float().t() + bias.float()
    out = ops.linear(a.to(dev), w.to(dev), bias.to(dev))
    check(out, ref, dtype, f"gemm {shape}")
    for tile in (1, 2, 3, 4, 5, 6):
        out = ops.linear(a.to(dev), w.to(dev), bias.to(dev), force_tile=tile)
        check(out, ref, dtype, f"gemm {shape} tile {tile}")
    if K >= 256:
        out = ops.linear(a.to(dev), w.to(dev), bias.to(dev), force_split_k=3, force_tile=2)
        check(out, ref, dtype, f"gemm {shape} split-K 3")


@pytest.mark.parametrize("shape", [(9000, 1000, 64), (9000, 1000, 200), (66000, 320, 320), (20000, 520, 128)])
def test_gemm_persistent_stream(shape):
    """More output tiles than co-resident blocks: every block walks several tiles and the K-tile stream (LDS-DMA
    prefetch) runs across tile boundaries, including K = one tile and ragged M / N / K, with residual + per-batch
    vector in the coalesced epilogue.  All tile configs (2- and 3-stage pipelines) must agree with fp32."""
    from theatergen_amd import ops
    dev = _dev()
    dtype = torch.bfloat16
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    a, w = rnd((M, K), dtype, g), rnd((N, K), dtype, g, 1 / math.sqrt(K))
    bias, res = rnd((N,), dtype, g), rnd((M, N), dtype, g)
    ad, wd, bd, rd = a.to(dev), w.to(dev), bias.to(dev), res.to(dev)
    ref = (ad.float() @ wd.float().t() + bd.float() + rd.float()).cpu()
    outs = []
    for tile in (0, 1, 2, 5, 6):
        out = ops.linear(ad, wd, bd, res=rd, force_tile=tile)
        check(out, ref, dtype, f"persistent gemm {shape} tile {tile}")
        outs.append(out)
    same = torch.equal(outs[0], outs[1])
    assert same
    # run-to-run determinism (no atomics, fixed tile walk)
    again = ops.linear(ad, wd, bd, res=rd)
    same = torch.equal(outs[0], again)
    assert same


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogues(dtype):
    from theatergen_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(7)
    B, rows, N, K = 3, 50, 192, 128
    M = B * rows
    a, w = rnd((M, K), dtype, g), rnd((N, K), dtype, g, 1 / math.sqrt(K))
    bias, bvec, res = rnd((N,), dtype, g), rnd((B, N), dtype, g), rnd((M, N), dtype, g)
    base = a.float() @ w.float().t() + bias.float() + bvec.float().repeat_interleave(rows, 0) + res.float()
    for act, fn in ((ops.ACT_NONE, lambda v: v), (ops.ACT_SILU, F.silu), (ops.ACT_GELU, F.gelu)):
        for split in (0, 2):
            out = ops.linear(a.to(dev), w.to(dev), bias.to(dev), bvec=bvec.to(dev), rows_per_batch=rows, res=res.to(dev),
                             act=act, out_scale=0.5, force_split_k=split)
            check(out, fn(base) * 0.5, dtype, f"epilogue act={act} split={split}")
    # two-source A (channel concat) + transposed tail columns (the V^T operand of attention)
    a0, a1 = rnd((M, 64), dtype, g), rnd((M, 128), dtype, g)
    w2 = rnd((N, 192), dtype, g, 1 / math.sqrt(192))
    ref = torch.cat([a0, a1], 1).float() @ w2.float().t()
    n_split = 128
    ldt = 56
    out = torch.zeros((M, n_split), dtype=dtype, device=dev)
    out_t = torch.zeros((B, N - n_split, ldt), dtype=dtype, device=dev)
    ops.gemm(a0.to(dev), w2.to(dev), M, N, 192, a1=a1.to(dev), c0=64, c1=128, rows_per_batch=rows, out=out,
             n_split=n_split, out_t=out_t, ldt=ldt)
    check(out, ref[:, :n_split], dtype, "two-source main")
    ref_t = ref[:, n_split:].reshape(B, rows, N - n_split).permute(0, 2, 1)
    check(out_t[:, :, :rows], ref_t, dtype, "transposed tail")
    assert (out_t[:, :, rows:] == 0).all()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(200, 1280, 320), (4096, 2560, 320), (130, 512, 64)])
def test_gemm_fused_geglu(dtype, shape):
    """FF1 + GEGLU in one launch (packed a|gate weight rows) == Linear -> chunk -> a * gelu(gate)."""
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import pack_geglu
    dev = _dev()
    M, N, K = shape
    g = torch.Generator().manual_seed(N + K)
    a, w, b = rnd((M, K), dtype, g), rnd((N, K), dtype, g, 1 / math.sqrt(K)), rnd((N,), dtype, g)
    y = a.float() @ w.float().t() + b.float()
    ref = y[:, :N // 2] * F.gelu(y[:, N // 2:])
    wp, bp = pack_geglu(w, b)
    out = ops.gemm(a.to(dev), wp.to(dev), M, N, K, bias=bp.to(dev), geglu=True)
    big = ops.gemm(a.to(dev), wp.to(dev), M, N, K, bias=bp.to(dev), geglu=True, force_tile=6)     # 256x256 tile, 8 waves of 128x64
    same = torch.equal(out, big)
    assert same
    assert out.shape == (M, N // 2)
    check(out, ref, dtype, f"fused geglu {shape}")


CONV_CASES = [
    # batch, h, w, cin, c1, cout, stride, upsample
    (2, 16, 16, 64, 0, 64, 1, False), (2, 16, 16, 128, 0, 64, 2, False), (1, 8, 8, 64, 0, 128, 1, True),
    (2, 8, 8, 128, 64, 128, 1, False), (2, 64, 64, 320, 0, 320, 1, False), (3, 5, 7, 64, 0, 64, 1, False),
    (2, 7, 7, 64, 0, 64, 2, False), (2, 8, 8, 1280, 1280, 1280, 1, False),
    # LDS-halo kernel (stride 1, width 16/32/64, M >= 4096): all three widths, two-source concat, ragged N
    (16, 16, 16, 128, 0, 192, 1, False), (4, 32, 32, 64, 64, 128, 1, False), (1, 64, 64, 64, 0, 320, 1, False),
    (2, 64, 64, 128, 64, 64, 1, False),
    # LDS-halo kernel on the nearest-x2 upsampled input (output widths 64 / 32 / 16)
    (1, 32, 32, 64, 0, 128, 1, True), (4, 16, 16, 128, 0, 64, 1, True), (16, 8, 8, 64, 0, 192, 1, True),
    # LDS-halo kernel at width 8: two whole 8x8 images per block (M >= 1024), two-source, ragged N
    (16, 8, 8, 128, 0, 192, 1, False), (32, 8, 8, 64, 64, 320, 1, False),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3(dtype, case):
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import pack_conv3x3
    dev = _dev()
    B, h, w, cin, c1, cout, stride, up = case
    g = torch.Generator().manual_seed(sum(case[:6]))
    ctot = cin + c1
    x = rnd((B, ctot, h, w), dtype, g)
    wt = rnd((cout, ctot, 3, 3), dtype, g, 1 / math.sqrt(9 * ctot))
    bias = rnd((cout,), dtype, g)
    xin = x.float()
    if up:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, wt.float(), bias.float(), stride=stride, padding=1)
    tok = x.permute(0, 2, 3, 1).reshape(B * h * w, ctot)
    x0 = tok[:, :cin].contiguous().to(dev)
    x1 = tok[:, cin:].contiguous().to(dev) if c1 else None
    out = ops.conv3x3(x0, pack_conv3x3(wt).to(dev), B, h, w, cin, x1=x1, c1=c1, stride=stride, upsample=up, bias=bias.to(dev))
    oh, ow = ref.shape[-2:]
    got = out.float().cpu().reshape(B, oh, ow, cout).permute(0, 3, 1, 2)
    check(got, ref, dtype, f"conv {case}")


@pytest.mark.parametrize("kind,shape", [
    ("gemm", (1000, 1252, 7680)),            # ragged M / N, 64x64 tiles, forced 4-way split of every tile
    ("gemm", (16384, 640, 2560)),            # 128x128 tiles, forced 3-way split
    ("conv", (16, 32, 32, 1280, 0, 640)),    # LDS-halo kernel: 640 tiles on 512 slots, 128 tail tiles cut over 64-channel chunks
    ("conv", (16, 16, 16, 1280, 1280, 1280)),  # LDS-halo kernel, two-source, 320 tiles all split
    ("conv", (16, 8, 8, 640, 640, 1280)),    # 8x8 weight-streaming layer (halo kernel, 2 images / block, two-source, all tiles split)
])
def test_gemm_tail_split(kind, shape, monkeypatch):
    """K-split of the tail tiles (grid rounds that would leave CUs idle): the heuristic must actually take the split
    path for these shapes (checked through the profiling records), partial sums + reduce must equal fp32, with
    bias + residual + per-batch vector in the reduce epilogue, and the result must be run-to-run identical."""
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import pack_conv3x3
    dev = _dev()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(sum(shape))
    monkeypatch.setenv("TG_GEMM_FLAGS", "128")     # the 32-wide case would go to the slab conv kernel (no K split there)
    if kind == "gemm":
        M, N, K = shape
        a = rnd((M, K), dtype, g).to(dev)
        w = rnd((N, K), dtype, g, 1 / math.sqrt(K)).to(dev)
        bias, res = rnd((N,), dtype, g).to(dev), rnd((M, N), dtype, g).to(dev)
        nb = 4 if M % 4 == 0 else 1
        bvec = rnd((nb, N), dtype, g).to(dev)
        ref = a.float() @ w.float().t() + bias.float() + res.float() + bvec.float().repeat_interleave(M // nb, dim=0)
        # (the measured cost model never splits a plain GEMM on its own any more: forced here to cover the reduce epilogue)
        fn = lambda: ops.linear(a, w, bias, res=res, bvec=bvec, rows_per_batch=M // nb, force_split_k=4 if M == 1000 else 3)
    else:
        B, h, wd, cin, c1, cout = shape
        ctot = cin + c1
        x = rnd((B, ctot, h, wd), dtype, g).to(dev)
        wt = rnd((cout, ctot, 3, 3), dtype, g, 1 / math.sqrt(9 * ctot)).to(dev)
        bias = rnd((cout,), dtype, g).to(dev)
        ref = F.conv2d(x.float(), wt.float(), bias.float(), padding=1).permute(0, 2, 3, 1).reshape(B * h * wd, cout)
        tok = x.permute(0, 2, 3, 1).reshape(B * h * wd, ctot)
        x0 = tok[:, :cin].contiguous()
        x1 = tok[:, cin:].contiguous() if c1 else None
        wp = pack_conv3x3(wt)
        fn = lambda: ops.conv3x3(x0, wp, B, h, wd, cin, x1=x1, c1=c1, bias=bias)
    ops.gemm_profile_start()
    out = fn()
    torch.cuda.synchronize()
    recs = ops.gemm_profile_stop()
    assert len(recs) == 1 and recs[0]["splits"] > 1, recs
    check(out, ref.cpu(), dtype, f"tail split {kind} {shape} x{recs[0]['splits']}")
    again = fn()
    same = torch.equal(out, again)
    assert same


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 64, 320, 0, True), (2, 256, 1280, 640, True), (1, 16, 64, 0, False), (3, 100, 128, 64, True),
                                  (2, 4096, 320, 0, True), (2, 64, 2560, 0, True),
                                  # one-launch small-map path (hw <= 256, channels per group a multiple of 8)
                                  (2, 64, 1280, 0, True), (2, 256, 1280, 1280, True), (3, 256, 1280, 0, False),
                                  (2, 100, 256, 0, True), (1, 16, 512, 256, True), (2, 250, 1024, 0, True)])
def test_groupnorm(dtype, case):
    from theatergen_amd import ops
    dev = _dev()
    B, hw, c0, c1, silu = case
    g = torch.Generator().manual_seed(hw + c0)
    C = c0 + c1
    x = rnd((B, hw, C), dtype, g) + 0.5
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).to(dtype), (0.1 * torch.randn(C, generator=g)).to(dtype)
    ref = F.group_norm(x.float().permute(0, 2, 1), 32, gamma.float(), beta.float(), 1e-5).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    x0 = x[:, :, :c0].reshape(B * hw, c0).contiguous().to(dev)
    x1 = x[:, :, c0:].reshape(B * hw, c1).contiguous().to(dev) if c1 else None
    out = ops.groupnorm(x0, B, hw, 32, 1e-5, gamma.to(dev), beta.to(dev), silu=silu, x1=x1)
    check(out.reshape(B, hw, C), ref, dtype, f"groupnorm {case}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(8192, 320), (300, 1280), (33, 768), (32, 2048), (5, 64)])
def test_layernorm(dtype, shape):
    from theatergen_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(shape[0])
    rows, C = shape
    x = rnd((rows, C), dtype, g) + 0.3
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).to(dtype), (0.1 * torch.randn(C, generator=g)).to(dtype)
    ref = F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), 1e-5)
    out = ops.layernorm(x.to(dev), gamma.to(dev), beta.to(dev))
    check(out, ref, dtype, f"layernorm {shape}")
    out = ops.layernorm(x.to(dev), None, None)
    check(out, F.layer_norm(x.float(), (C,)), dtype, f"layernorm no-affine {shape}")


def _attn_ref(q, k, v, heads, scale):
    B, N, Cc = q.shape
    d = Cc // heads
    qh = q.float().reshape(B, N, heads, d).permute(0, 2, 1, 3)
    kh = k.float().reshape(B, -1, heads, d).permute(0, 2, 1, 3)
    vh = v.float().reshape(B, -1, heads, d).permute(0, 2, 1, 3)
    p = (qh @ kh.transpose(-1, -2) * scale).softmax(-1)
    return (p @ vh).permute(0, 2, 1, 3).reshape(B, N, Cc), p


ATTN_CASES = [
    # heads, d, n_q, len0, len1
    (8, 40, 256, 256, 0), (8, 40, 200, 77, 4), (8, 80, 64, 77, 16), (8, 160, 64, 64, 0), (8, 160, 36, 77, 4),
    (10, 64, 144, 144, 0), (20, 64, 96, 77, 16), (2, 32, 70, 70, 0), (4, 128, 16, 81, 0), (4, 16, 8, 41, 0),
    (8, 40, 1024, 1024, 0), (12, 64, 16, 273, 0),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention(dtype, case):
    from theatergen_amd import ops
    dev = _dev()
    heads, d, n_q, len0, len1 = case
    B = 2
    Cc = heads * d
    g = torch.Generator().manual_seed(sum(case))
    q = rnd((B, n_q, Cc), dtype, g)
    k0, v0 = rnd((B, len0, Cc), dtype, g), rnd((B, len0, Cc), dtype, g)
    scale = d ** -0.5
    ref, _ = _attn_ref(q, k0, v0, heads, scale)
    w1 = 0.4
    pad0 = (len0 + 7) // 8 * 8
    # V^T buffers: padding columns deliberately poisoned with NaN bits: the kernel must not read them as data
    vt0 = torch.full((B, Cc, pad0), float("nan"), dtype=dtype)
    vt0[:, :, :len0] = v0.permute(0, 2, 1)
    args = dict(k1=None, vt1=None, len1=0, w1=0.0)
    if len1:
        k1, v1 = rnd((B, len1, Cc), dtype, g), rnd((B, len1, Cc), dtype, g)
        ref = ref + w1 * _attn_ref(q, k1, v1, heads, scale)[0]
        pad1 = (len1 + 7) // 8 * 8
        vt1 = torch.full((B, Cc, pad1), float("nan"), dtype=dtype)
        vt1[:, :, :len1] = v1.permute(0, 2, 1)
        k1d, vt1d = k1.to(dev), vt1.to(dev)
        args = dict(k1=k1d, k1_ld=Cc, k1_bs=len1 * Cc, vt1=vt1d, vt1_ld=pad1, vt1_bs=Cc * pad1, len1=len1, w1=w1)
    out = torch.zeros((B, n_q, Cc), dtype=dtype, device=dev)
    ops.attention(q.to(dev), Cc, n_q * Cc, k0.to(dev), Cc, len0 * Cc, vt0.to(dev), pad0, Cc * pad0, len0, B, heads, d, n_q,
                  scale, out, Cc, n_q * Cc, **args)
    check(out, ref, dtype, f"attention {case}", scale=1.5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_forced_rescale(dtype):
    """Online-softmax rescale branch: one key far above the rest in a LATE tile (cdna guide rule 26)."""
    from theatergen_amd import ops
    dev = _dev()
    heads, d, n_q, len0, B = 2, 64, 64, 320, 1
    Cc = heads * d
    g = torch.Generator().manual_seed(99)
    q, k0, v0 = rnd((B, n_q, Cc), dtype, g), rnd((B, len0, Cc), dtype, g), rnd((B, len0, Cc), dtype, g)
    k0[0, 300] = (q[0, 5].float() * 3).to(dtype)      # spike for query 5 in tile 4
    k0[0, 10] = (q[0, 40].float() * 2).to(dtype)      # and an early one for query 40
    ref, _ = _attn_ref(q, k0, v0, heads, d ** -0.5)
    vt0 = v0.permute(0, 2, 1).contiguous()
    out = torch.zeros((B, n_q, Cc), dtype=dtype, device=dev)
    ops.attention(q.to(dev), Cc, n_q * Cc, k0.to(dev), Cc, len0 * Cc, vt0.to(dev), len0, Cc * len0, len0, B, heads, d, n_q,
                  d ** -0.5, out, Cc, n_q * Cc)
    check(out, ref, dtype, "attention forced rescale", scale=1.5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attn_probs(dtype):
    from theatergen_amd import ops
    dev = _dev()
    heads, d, n_q, ln, B = 8, 40, 64, 77, 2
    Cc = heads * d
    g = torch.Generator().manual_seed(5)
    q, k = rnd((B, n_q, Cc), dtype, g), rnd((B, ln, Cc), dtype, g)
    _, p = _attn_ref(q, k, k, heads, d ** -0.5)
    got = ops.attn_probs(q.to(dev), Cc, n_q * Cc, k.to(dev), Cc, ln * Cc, B, 0, heads, d, n_q, ln, d ** -0.5)
    assert torch.allclose(got.cpu(), p, rtol=2e-3, atol=2e-5)
    tok = torch.tensor([1, 3, 70], dtype=torch.int32, device=dev)
    got = ops.attn_probs(q.to(dev), Cc, n_q * Cc, k.to(dev), Cc, ln * Cc, B, 1, heads, d, n_q, ln, d ** -0.5, tokens=tok)
    assert torch.allclose(got.cpu(), p[1:, :, :, [1, 3, 70]], rtol=2e-3, atol=2e-5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_elementwise_and_boundary_convs(dtype):
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import pack_conv3x3
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    x = rnd((300, 512), dtype, g)
    ref = x[:, :256].float() * F.gelu(x[:, 256:].float())
    check(ops.geglu(x.to(dev)), ref, dtype, "geglu")
    check(ops.act(x.to(dev), ops.ACT_SILU), F.silu(x.float()), dtype, "silu")
    y = rnd((300, 512), dtype, g)
    check(ops.add(x.to(dev), y.to(dev)), x.float() + y.float(), dtype, "add")
    # conv_in: NCHW fp32 sample -> token-major
    s = torch.randn(2, 4, 16, 16, generator=g)
    w, b = rnd((64, 4, 3, 3), dtype, g, 1 / 6), rnd((64,), dtype, g)
    ref = F.conv2d(s, w.float(), b.float(), padding=1).permute(0, 2, 3, 1).reshape(-1, 64)
    check(ops.conv_in(s.to(dev), pack_conv3x3(w).to(dev), b.to(dev), 64, dtype), ref, dtype, "conv_in")
    # conv_out: token-major -> NCHW fp32
    xin = rnd((2, 64, 16, 16), dtype, g)
    w, b = rnd((4, 64, 3, 3), dtype, g, 1 / 24), rnd((4,), dtype, g)
    ref = F.conv2d(xin.float(), w.float(), b.float(), padding=1)
    tok = xin.permute(0, 2, 3, 1).reshape(-1, 64).contiguous()
    got = ops.conv_out(tok.to(dev), pack_conv3x3(w).to(dev), b.to(dev), 2, 16, 16, 4, torch.float32)
    assert torch.allclose(got.cpu(), ref, rtol=1e-3, atol=1e-3)
    # timestep embedding
    from oracle.unet import timestep_sinusoid
    t = torch.tensor([981.0, 1.0, 500.0])
    got = ops.timestep_embedding(t.to(dev), 3, 320, True, 0.0, dtype, t_stride=1)
    check(got, timestep_sinusoid(t, 320, True, 0.0), dtype, "timestep embedding")


def test_step_epilogue_and_latent_ops():
    """fp32 paths: tolerance 1e-5 relative (fma contraction differences only)."""
    from oracle import ddim as oddim
    from oracle import box_geometry as geo
    from oracle import latent_ops as ol
    from theatergen_amd import ops
    from theatergen_amd.scheduler import DDIMScheduler
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    for pred in ("epsilon", "v_prediction"):
        osch = oddim.DDIMSchedule(prediction_type=pred)
        osch.set_timesteps(50)
        sch = DDIMScheduler(prediction_type=pred)
        sch.set_timesteps(50)
        coef = sch.coef_table().to(dev)
        lat = torch.randn(2, 4, 64, 64, generator=g)
        frozen = torch.randn(51, 2, 4, 64, 64, generator=g)
        mask = (torch.rand(64, 64, generator=g) > 0.5).float()
        hist = torch.zeros(51, 2, 4, 64, 64, device=dev)
        lat_d = lat.clone().to(dev)
        step = torch.zeros(1, dtype=torch.int32, device=dev)
        ref = lat.clone()
        model_in = torch.zeros(4, 4, 64, 64, dtype=torch.bfloat16, device=dev)
        for i in range(3):
            npred = torch.randn(4, 4, 64, 64, generator=g)
            t = int(osch.timesteps[i])
            ref = oddim.step_epilogue(osch, npred, t, ref, 7.5, frozen[i + 1] if i < 2 else None, mask)
            ops.step_epilogue(npred.to(dev), lat_d, 7.5, coef, step, prediction_type=0 if pred == "epsilon" else 1,
                              frozen=frozen.to(dev), frozen_mask=mask.to(dev), frozen_steps=2, history=hist, model_in=model_in)
            assert torch.allclose(lat_d.cpu(), ref, rtol=1e-5, atol=1e-5), f"step {i} {pred}"
            assert torch.equal(hist[i + 1], lat_d)
            assert torch.equal(model_in[:2].float(), lat_d.to(torch.bfloat16).float())
        assert int(step.item()) == 3
    # blend / shift / compose
    bg, fg = torch.randn(1, 4, 64, 64, generator=g), torch.randn(1, 4, 64, 64, generator=g)
    m = geo.proportion_to_mask([0.1, 0.2, 0.6, 0.9], 64, 64)
    got = ops.blend_latents(bg.to(dev), fg.to(dev), m.to(dev), 0.01)
    assert torch.allclose(got.cpu(), ol.blend_latents(bg, fg, m, 0.01), rtol=1e-6, atol=1e-6)
    t = torch.randn(5, 1, 4, 64, 64, generator=g)
    for dx, dy in ((8, -16), (-24, 0), (0, 0), (64, 8)):
        assert torch.equal(ops.shift(t.to(dev), dx, dy).cpu(), geo.shift_tensor(t, dx, dy))
    dst, src = torch.randn(51, 1, 4, 64, 64, generator=g), torch.randn(51, 1, 4, 64, 64, generator=g)
    want = dst * (1 - m) + src * m
    got = ops.masked_compose_(dst.clone().to(dev), src.to(dev), m.to(dev))
    assert torch.allclose(got.cpu(), want, rtol=1e-6, atol=1e-6)


def test_guidance_reductions():
    from oracle import guidance_loss as og
    from tests.golden import gen_common as gc
    from theatergen_amd import guidance as G
    dev = _dev()
    keys = gc.GUIDANCE_KEYS
    for nbox in (1, 2, 4):
        maps, _ = gc.guidance_attn_maps(nbox)
        for kw in (dict(use_ratio_based_loss=False, fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=4.0),
                   dict(use_ratio_based_loss=True)):
            saved = {k: v.clone().requires_grad_(True) for k, v in maps.items()}
            ref = og.compute_ca_lossv3(saved, gc.GUIDANCE_BOXES[nbox], gc.GUIDANCE_POSITIONS[nbox], keys, **kw)
            ref_grads = torch.autograd.grad(ref, [saved[k] for k in keys])
            dmaps = {k: v.to(dev) for k, v in maps.items()}
            loss, grads = G.compute_ca_lossv3(dmaps, gc.GUIDANCE_BOXES[nbox], gc.GUIDANCE_POSITIONS[nbox], keys,
                                              return_grads=True, **kw)
            assert abs(loss.item() - ref.item()) <= 2e-5 * max(1.0, abs(ref.item())), (nbox, kw, loss.item(), ref.item())
            for k, rg in zip(keys, ref_grads):
                assert torch.allclose(grads[k].cpu(), rg, rtol=1e-4, atol=1e-7), (nbox, k)


def test_guidance_reference_attention_transfer():
    """compute_ca_lossv3 with ref_ca_saved_attns (reference utils/guidance.py:150-242): value and d loss / d A vs the
    oracle (itself pinned on the imported reference's ``guid.2.withref.loss``)."""
    from oracle import guidance_loss as og
    from tests.golden import gen_common as gc
    from theatergen_amd import guidance as G
    dev = _dev()
    keys = gc.GUIDANCE_KEYS
    g = torch.Generator().manual_seed(77)
    maps, _ = gc.guidance_attn_maps(2)
    refs = gc.guidance_ref_maps(g)
    for kw in (dict(ref_ca_last_token_only=True), dict(ref_ca_last_token_only=False),
               dict(ref_ca_word_token_only=True, word_token_indices=[3, 7])):
        args = dict(ref_ca_saved_attns=refs, index=3, ref_ca_loss_weight=2.0, use_ratio_based_loss=False, **kw)
        saved = {k: v.clone().requires_grad_(True) for k, v in maps.items()}
        want = og.compute_ca_lossv3(saved, gc.GUIDANCE_BOXES[2], gc.GUIDANCE_POSITIONS[2], keys, **args)
        want_grads = torch.autograd.grad(want, [saved[k] for k in keys])
        dmaps = {k: v.to(dev) for k, v in maps.items()}
        loss, grads = G.compute_ca_lossv3(dmaps, gc.GUIDANCE_BOXES[2], gc.GUIDANCE_POSITIONS[2], keys, return_grads=True,
                                          **args)
        assert abs(loss.item() - want.item()) <= 2e-5 * max(1.0, abs(want.item())), (kw, loss.item(), want.item())
        for k, rg in zip(keys, want_grads):
            assert torch.allclose(grads[k].cpu(), rg, rtol=1e-4, atol=1e-6), (kw, k)
        plain = G.compute_ca_lossv3(dmaps, gc.GUIDANCE_BOXES[2], gc.GUIDANCE_POSITIONS[2], keys, **args)
        assert abs(plain.item() - loss.item()) <= 1e-6 * max(1.0, abs(loss.item()))


def test_guidance_at_sd21_map_sizes_and_batched_launch():
    """BASELINE.json configs[3] (SD-2.1 768x768, 4 boxes): the guidance keys are the mid block (12 x 12 = 144 pixels) and the
    three up-1 layers (24 x 24 = 576), 20 heads; non-power-of-two maps, 4 boxes incl. a two-box object, top-k and ratio
    forms, loss and d loss / d A against the oracle.  compute_ca_lossv3 issues ONE batched launch for all its terms: the
    result must equal the per-term launches bit for bit."""
    from oracle import guidance_loss as og
    from tests.golden import gen_common as gc
    from theatergen_amd import guidance as G
    from theatergen_amd import ops
    dev = _dev()
    keys = gc.GUIDANCE_KEYS
    hw = {keys[0]: 144, keys[1]: 576, keys[2]: 576, keys[3]: 576}
    g = torch.Generator().manual_seed(4242)
    maps = {}
    for k in keys:
        a = torch.rand(1, 20, hw[k], 77, generator=g)
        maps[k] = a / a.sum(-1, keepdim=True)
    boxes, pos = gc.GUIDANCE_BOXES[4], gc.GUIDANCE_POSITIONS[4]
    for kw in (dict(use_ratio_based_loss=False, fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=4.0), dict(use_ratio_based_loss=True)):
        saved = {k: v.clone().requires_grad_(True) for k, v in maps.items()}
        ref = og.compute_ca_lossv3(saved, boxes, pos, keys, **kw)
        ref_grads = torch.autograd.grad(ref, [saved[k] for k in keys])
        dmaps = {k: v.to(dev) for k, v in maps.items()}
        loss, grads = G.compute_ca_lossv3(dmaps, boxes, pos, keys, return_grads=True, **kw)
        assert abs(loss.item() - ref.item()) <= 2e-5 * max(1.0, abs(ref.item())), (kw, loss.item(), ref.item())
        for k, rg in zip(keys, ref_grads):
            assert torch.allclose(grads[k].cpu(), rg, rtol=1e-4, atol=1e-7), (kw, k)
        # the same terms, one launch each (the round-1 path): identical bits
        seq = torch.zeros(1, dtype=torch.float32, device=dev)
        seq_grads = {k: torch.zeros_like(dmaps[k][0]) for k in keys}
        norm = 1.0 / (len(boxes) * len(keys))
        for k in keys:
            G.add_ca_loss_per_attn_map_to_loss(seq, dmaps[k][0].contiguous(), len(boxes), boxes, pos, grad=seq_grads[k], scale=norm, **kw)
        assert seq.item() == loss.item(), (seq.item(), loss.item())
        for k in keys:
            assert torch.equal(seq_grads[k], grads[k][0])
    # a map too large for the LDS-resident select is an argument error, not a launch failure (hw = 9216 = SD-2.1 level 0)
    big = torch.rand(2, 9216, 8, device=dev)
    with pytest.raises(RuntimeError, match="too large"):
        ops.guidance_topk(big, 1, torch.ones(9216, device=dev), 10, 10, 1.0, 1.0, 1.0, torch.zeros(1, device=dev))
    b = ops.GuidanceBatch(dev)
    b.add(b.KIND_TOPK, big, 1, torch.ones(9216, device=dev), 1.0, k_fg=10, k_bg=10, fg_w=1.0, bg_w=1.0)
    with pytest.raises(RuntimeError, match="too large"):
        b.flush(torch.zeros(1, device=dev))


def test_latent_shift_and_compose_at_96x96():
    """configs[3] geometry: 768 x 768 -> 96 x 96 latents, 4 objects; zero-filled shift and masked paste are data movement:
    bit-exact against the oracle."""
    from oracle import box_geometry as geo
    from oracle import latent_ops as ol
    from theatergen_amd import latents as L
    from theatergen_amd import utils as U
    dev = _dev()
    g = torch.Generator().manual_seed(96)
    S = 11
    lat_all = [torch.randn(S, 1, 4, 96, 96, generator=g) for _ in range(4)]
    masks = []
    for i in range(4):
        m = torch.zeros(96, 96, dtype=torch.bool)
        y0, x0 = 6 + 17 * i, 9 + 13 * i
        m[y0:y0 + 22 + 2 * i, x0:x0 + 18 + 3 * i] = True
        m &= torch.rand(96, 96, generator=g) > 0.1
        masks.append(m)
    boxes = [[0.05, 0.1, 0.35, 0.5], [0.5, 0.1, 0.9, 0.45], [0.1, 0.55, 0.45, 0.95], [0.55, 0.6, 0.95, 0.98]]
    for xo, yo in ((0.13, -0.21), (-0.5, 0.0), (0.07, 0.06)):
        got = U.shift_tensor(lat_all[0].to(dev), xo, yo, offset_normalized=True).cpu()
        assert torch.equal(got, geo.shift_tensor(lat_all[0], xo, yo, offset_normalized=True))
    new_l, new_m, offs = L.align_with_bboxes([x.to(dev) for x in lat_all], masks, boxes)
    rl, rm, roffs = ol.align_with_bboxes(lat_all, masks, boxes)
    assert offs == roffs and all(torch.equal(a, b) for a, b in zip(new_m, rm))
    assert all(torch.equal(a.cpu(), b) for a, b in zip(new_l, rl))
    bg = torch.randn(1, 4, 96, 96, generator=g)

    class _Cfg:
        in_channels = 4

    class _Unet:
        config = _Cfg()
        dtype = torch.float32

    ad = type("A", (), {"pipe": type("P", (), {"unet": _Unet(), "scheduler": type("Sc", (), {"init_noise_sigma": 1.0})()})()})()
    comp, fgidx = L.compose_latents(ad, None, new_l, new_m, S - 1, 1, 768, 768, latents_bg=bg.to(dev))
    rcomp, rfg = ol.compose_latents(rl, rm, bg, S)
    assert torch.equal(fgidx.cpu(), rfg) and torch.equal(comp.cpu(), rcomp)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(66000, 320, 320), (9000, 1000, 192), (4096, 1280, 1280), (300, 640, 64), (16384, 640, 2560)])
def test_gemm_big_tiles(dtype, shape):
    """The big-tile kernel (tg_gemm_bt.hip: force_tile 10 = 256 x 256 — its 128 x 320 sibling was removed in round 5; 8 waves, persistent, one barrier per
    K-tile, asm LDS-DMA / fragment reads with counted waits): ragged M and N, K of one tile up to 40 tiles, several output
    tiles per workgroup (the cross-tile prefetch), bias + per-batch vector + residual in the chunked LDS epilogue.  They
    accumulate in the same order as the 128 x 128 kernel, so the results must also be BIT-identical to it."""
    from theatergen_amd import ops
    dev = _dev()
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    rows = 100 if M % 100 == 0 else M
    a, w = rnd((M, K), dtype, g), rnd((N, K), dtype, g, 1 / math.sqrt(K))
    bias, res, bvec = rnd((N,), dtype, g), rnd((M, N), dtype, g), rnd((M // rows, N), dtype, g)
    ad, wd, bd, rd, vd = a.to(dev), w.to(dev), bias.to(dev), res.to(dev), bvec.to(dev)
    ref = (ad.float() @ wd.float().t() + bd.float() + rd.float() + vd.float().repeat_interleave(rows, 0)).cpu()
    base = ops.linear(ad, wd, bd, res=rd, bvec=vd, rows_per_batch=rows, force_tile=1)
    for tile in (10,):
        out = ops.linear(ad, wd, bd, res=rd, bvec=vd, rows_per_batch=rows, force_tile=tile)
        check(out, ref, dtype, f"big tile {tile} {shape}")
        same = torch.equal(out, base)
        assert same, f"big tile {tile} {shape}: not bit-identical to the 128 x 128 kernel"
        again = ops.linear(ad, wd, bd, res=rd, bvec=vd, rows_per_batch=rows, force_tile=tile)
        same = torch.equal(out, again)
        assert same
    # activation epilogue + scale
    out = ops.linear(ad, wd, bd, act=ops.ACT_SILU, out_scale=0.5, force_tile=10)
    check(out, F.silu(ad.float() @ wd.float().t() + bd.float()).cpu() * 0.5, dtype, f"big tile silu {shape}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_big_tiles_qkv_split_and_geglu(dtype):
    """Big tiles with the attention operand layout (Q | K token-major + V^T per batch item, n_split on a wave-tile boundary)
    and the fused GEGLU epilogue (256 x 256 tile)."""
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import pack_geglu
    dev = _dev()
    g = torch.Generator().manual_seed(99)
    B, rows, C = 3, 1024, 320
    M, N, K = B * rows, 3 * C, C
    a, w = rnd((M, K), dtype, g), rnd((N, K), dtype, g, 1 / math.sqrt(K))
    ref = a.float() @ w.float().t()
    for tile in (1, 10):
        out = torch.zeros((M, 2 * C), dtype=dtype, device=dev)
        out_t = torch.zeros((B, C, rows), dtype=dtype, device=dev)
        ops.gemm(a.to(dev), w.to(dev), M, N, K, rows_per_batch=rows, out=out, n_split=2 * C, out_t=out_t, ldt=rows, force_tile=tile)
        check(out, ref[:, :2 * C], dtype, f"qkv main tile {tile}")
        check(out_t, ref[:, 2 * C:].reshape(B, rows, C).permute(0, 2, 1), dtype, f"qkv V^T tile {tile}")
    M2, N2, K2 = 4096, 2560, 320
    a2, w2, b2 = rnd((M2, K2), dtype, g), rnd((N2, K2), dtype, g, 1 / math.sqrt(K2)), rnd((N2,), dtype, g)
    y = a2.float() @ w2.float().t() + b2.float()
    wp, bp = pack_geglu(w2, b2)
    small = ops.gemm(a2.to(dev), wp.to(dev), M2, N2, K2, bias=bp.to(dev), geglu=True, force_tile=1)
    big = ops.gemm(a2.to(dev), wp.to(dev), M2, N2, K2, bias=bp.to(dev), geglu=True, force_tile=10)
    check(big, y[:, :N2 // 2] * F.gelu(y[:, N2 // 2:]), dtype, "geglu big tile")
    same = torch.equal(small, big)
    assert same


# ---- slab conv kernel (tg_conv_slab.hip): BM x 320 tiles, window staged once per 320 channels, GroupNorm + SiLU prologue ----
SLAB_CASES = [
    # B, h, w, cin, c1, cout, force_tile (11 = slab kernel regardless of the tile count, 12 = channel chunks split in two)
    (5, 16, 16, 1280, 0, 1280, 12),     # 16-wide maps: a 32-pixel fragment spans two image rows; 20 chunks in 10 + 10
    (4, 16, 16, 1280, 640, 1280, 12),   # two sources, 30 chunks
    (16, 16, 16, 640, 0, 1280, 0),      # the heuristic splits the SD-1.5 16x16 level (128 tiles) in two
    (2, 64, 64, 192, 0, 320, 12),       # 3 chunks in 2 + 1
    (2, 64, 64, 320, 0, 320, 11),       # SD-1.5 level 0; 64 tiles: fewer tiles than workgroup slots
    (2, 64, 64, 640, 320, 320, 11),     # up block 3: two sources (hidden + skip), 15 channel chunks
    (1, 128, 64, 64, 0, 320, 11),       # ONE channel chunk (no staging under the loop), tall map
    (3, 32, 32, 320, 0, 640, 11),       # two N tiles share a window
    (3, 32, 32, 1280, 640, 640, 11),    # SD-1.5 up block 2, 30 chunks
    (2, 64, 32, 128, 0, 320, 11),       # two chunks: the look-ahead request never fires
    (34, 32, 32, 320, 0, 320, 11),      # 272 tiles on 256 persistent workgroups: second, partial round
    # round 3, PATCH tiles: the image is wider than the tile (128-wide maps as 2 x 64 patches, 96-wide maps as 4 x 32 patches): halo
    # columns are real neighbour pixels, a wave's pixels are not one contiguous token range
    (1, 8, 128, 128, 0, 320, 11),       # SDXL level-0 width, 8 tiles in 4 patch rows x 2 patch columns
    (2, 4, 128, 320, 64, 640, 11),      # two images, two sources, two N tiles
    (1, 8, 96, 192, 0, 320, 11),        # SD-2.1 level-0 width: 3 patch columns x 2 patch rows of 4 x 32
    (2, 12, 96, 320, 320, 320, 11),     # two images x (3 x 3) patches, two sources
    (2, 128, 128, 320, 0, 320, 0),      # BASELINE configs[4] level 0 at batch 2: 256 tiles, taken by the heuristic
    (2, 96, 96, 320, 0, 320, 0),        # BASELINE configs[3] level 0 at batch 2: 144 tiles, taken by the heuristic (no 128-pixel halo fallback)
    # round 3, second pass: generic patches (8th entry = K splits over the channel chunks: force_split_k with force_tile 11, or the
    # heuristic's expected choice with force_tile 0).  16-pixel patch rows (48 = 3 x 16, 80 = 5 x 16): a 32-row MFMA block is two patch rows;
    # two 8 x 8 patches per tile (24 = 3 x 8, the 8 x 8 level): a tile may straddle two images; split tiles: fp32 partials in tile-local
    # row order, the reduce kernel maps rows to tokens
    (2, 48, 48, 128, 0, 320, 11),       # SD-2.1 level-1 width: 3 patch columns x 6 patch rows per image
    (2, 48, 48, 640, 320, 640, 11, 3),  # two sources, two N tiles, 15 chunks in 5 + 5 + 5
    (1, 16, 80, 64, 0, 320, 11),        # 5 patch columns, one chunk
    (2, 24, 24, 128, 0, 320, 11),       # SD-2.1 level-2 width: 9 patches per image, tile 4 holds the last patch of image 0 and the first of image 1
    (2, 24, 24, 1280, 0, 1280, 11, 5),  # 20 chunks in 5 x 4
    (4, 8, 8, 192, 64, 320, 11),        # the 8 x 8 level: two whole images per tile, two sources
    (16, 8, 8, 1280, 0, 1280, 11, 7),   # SD-1.5 8 x 8 level at CFG batch 16, 20 chunks in 6 x 3 + 2
    (2, 8, 128, 128, 0, 320, 11, 2),    # 64-pixel patch rows, split
    (2, 16, 96, 192, 0, 320, 11, 3),    # 32-pixel patch rows, split 1 + 1 + 1
    (6, 16, 16, 320, 0, 640, 11, 5),    # whole-row tiles with more than two splits
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", SLAB_CASES)
def test_conv_slab_kernel(dtype, case):
    """slab kernel vs fp32 F.conv2d, bias + per-image vector + residual + scale epilogue; and bit-identical to the LDS-halo
    kernel (same K order: channel chunk, tap, k) where that one takes the problem unsplit."""
    import os
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import pack_conv3x3
    dev = _dev()
    B, h, w, cin, c1, cout, ft = case[:7]
    fs = case[7] if len(case) > 7 else 0
    g = torch.Generator().manual_seed(sum(case))
    ctot = cin + c1
    x = rnd((B, ctot, h, w), dtype, g)
    wt = rnd((cout, ctot, 3, 3), dtype, g, 1 / math.sqrt(9 * ctot))
    bias, bvec, res = rnd((cout,), dtype, g), rnd((B, cout), dtype, g), rnd((B, cout, h, w), dtype, g)
    ref = (F.conv2d(x.float(), wt.float(), bias.float(), padding=1) + bvec.float()[:, :, None, None] + res.float()) * 0.5
    tok = x.permute(0, 2, 3, 1).reshape(B * h * w, ctot)
    x0 = tok[:, :cin].contiguous().to(dev)
    x1 = tok[:, cin:].contiguous().to(dev) if c1 else None
    kw = dict(x1=x1, c1=c1, bias=bias.to(dev), bvec=bvec.to(dev), rows_per_batch=h * w,
              res=res.permute(0, 2, 3, 1).reshape(B * h * w, cout).contiguous().to(dev), out_scale=0.5)
    wp = pack_conv3x3(wt).to(dev)
    fkw = dict(force_tile=ft, force_split_k=fs if ft == 11 else 0)
    plan = ops.conv3x3(x0, wp, B, h, w, cin, plan_only=True, **fkw, **kw)
    assert plan[3] == 4 and plan[2] == (fs if fs else (2 if ft == 12 or case[0] == 16 else 1))
    out = ops.conv3x3(x0, wp, B, h, w, cin, **fkw, **kw)
    got = out.float().cpu().reshape(B, h, w, cout).permute(0, 3, 1, 2)
    check(got, ref, dtype, f"slab conv {case}")
    # round 6: whole-row tiles of the 64 / 32 / 16-wide maps run on the ping-pong compute waves (tg_conv_slab_pp.hip, 16 x 16 x 32 MFMAs: another summation
    # order inside a K-step); TG_SLAB_PP=0 keeps conv_slab_kernel, whose outputs are bit-identical to the LDS-halo kernel's
    old_pp = os.environ.get("TG_SLAB_PP")
    os.environ["TG_SLAB_PP"] = "0"
    try:
        out1 = ops.conv3x3(x0, wp, B, h, w, cin, **fkw, **kw)
    finally:
        if old_pp is None:
            del os.environ["TG_SLAB_PP"]
        else:
            os.environ["TG_SLAB_PP"] = old_pp
    check(out1.float().cpu().reshape(B, h, w, cout).permute(0, 3, 1, 2), ref, dtype, f"slab conv (one compute wave per SIMD) {case}")
    pp_took_it = w in (16, 32, 64) and h % (128 // w) == 0  # default TG_SLAB_PP = 2: whole-row tiles of the 64 / 32 / 16-wide maps
    if not pp_took_it:
        assert torch.equal(out, out1), "patch tiles do not run on the ping-pong kernel"
    else:
        check(out.float(), out1.float(), dtype, f"ping-pong slab vs one-wave slab {case}", scale=1.0)
        again = ops.conv3x3(x0, wp, B, h, w, cin, **fkw, **kw)
        assert torch.equal(out, again), "the ping-pong slab kernel is deterministic"
    old = os.environ.get("TG_GEMM_FLAGS")
    os.environ["TG_GEMM_FLAGS"] = "128"                     # dev flag: the planner skips the slab kernel
    try:
        tm, tn, sp, kk = ops.conv3x3(x0, wp, B, h, w, cin, plan_only=True, **kw)
        assert kk != 4
        if kk == 2 and sp == 1 and plan[2] == 1:
            halo = ops.conv3x3(x0, wp, B, h, w, cin, **kw)
            assert torch.equal(out1, halo), "conv_slab_kernel and the halo kernel accumulate in the same order: outputs must be bit-identical"
    finally:
        if old is None:
            del os.environ["TG_GEMM_FLAGS"]
        else:
            os.environ["TG_GEMM_FLAGS"] = old


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 64, 64, 320, 0, 320, 11), (2, 64, 64, 640, 320, 320, 11), (3, 32, 32, 960, 320, 640, 11),
                                  (1, 64, 64, 64, 0, 320, 11), (2, 32, 32, 128, 0, 320, 11), (16, 32, 32, 640, 0, 640, 0),
                                  (16, 16, 16, 1280, 1280, 1280, 0), (3, 16, 16, 640, 0, 1280, 12),
                                  (2, 8, 128, 320, 0, 320, 11), (2, 12, 96, 640, 320, 320, 11), (2, 96, 96, 320, 0, 320, 0),
                                  # generic patches: 3 x 16 patch columns; two 8 x 8 patches per tile of DIFFERENT images (two coefficient
                                  # sets per tile); the 8 x 8 level; split tiles
                                  (2, 48, 48, 320, 0, 640, 11), (2, 24, 24, 640, 640, 1280, 11), (4, 8, 8, 128, 0, 320, 11),
                                  (2, 24, 24, 1280, 0, 1280, 11, 5), (2, 48, 48, 640, 0, 640, 11, 3)])
def test_conv_slab_groupnorm_prologue(dtype, case):
    """GroupNorm + SiLU applied while the window is staged == tg_groupnorm followed by the plain conv, bit for bit; the
    coefficients against an fp32 reference; the heuristic (force_tile 0) takes a layer that fills the chip."""
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import pack_conv3x3
    dev = _dev()
    B, h, w, cin, c1, cout, ft = case[:7]
    fs = case[7] if len(case) > 7 else 0
    g = torch.Generator().manual_seed(sum(case) + 1)
    ctot = cin + c1
    x = rnd((B, ctot, h, w), dtype, g) * 1.5 + 0.3
    wt = rnd((cout, ctot, 3, 3), dtype, g, 1 / math.sqrt(9 * ctot))
    bias = rnd((cout,), dtype, g)
    gamma, beta = (1 + 0.2 * torch.randn(ctot, generator=g)).to(dtype), (0.1 * torch.randn(ctot, generator=g)).to(dtype)
    tok = x.permute(0, 2, 3, 1).reshape(B * h * w, ctot)
    x0 = tok[:, :cin].contiguous().to(dev)
    x1 = tok[:, cin:].contiguous().to(dev) if c1 else None
    wp = pack_conv3x3(wt).to(dev)
    if ft == 0:
        assert ops.conv3x3_takes_gn(dtype, B, h, w, cin, c1, cout)
    coef = ops.groupnorm_coef(x0, B, h * w, 32, 1e-5, gamma.to(dev), beta.to(dev), x1=x1)
    xf = x.float().reshape(B, 32, -1)
    mean, var = xf.mean(-1), xf.var(-1, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    a_ref = rstd.repeat_interleave(ctot // 32, dim=1) * gamma.float()[None]
    d_ref = beta.float()[None] - mean.repeat_interleave(ctot // 32, dim=1) * a_ref
    cf = coef.cpu()
    assert torch.allclose(cf[:, 0], a_ref, rtol=2e-4, atol=1e-5) and torch.allclose(cf[:, 1], d_ref, rtol=2e-4, atol=2e-4)
    fused = ops.conv3x3(x0, wp, B, h, w, cin, x1=x1, c1=c1, bias=bias.to(dev), a_coef=coef, a_silu=True, force_tile=ft, force_split_k=fs)
    hn = ops.groupnorm(x0, B, h * w, 32, 1e-5, gamma.to(dev), beta.to(dev), silu=True, x1=x1)
    plain = ops.conv3x3(hn, wp, B, h, w, ctot, bias=bias.to(dev), force_tile=ft, force_split_k=fs)
    assert torch.equal(fused, plain), f"fused GroupNorm prologue differs from norm -> conv: {(fused.float() - plain.float()).abs().max().item()}"
    ref = F.conv2d(F.silu(F.group_norm(x.float(), 32, gamma.float(), beta.float(), 1e-5)), wt.float(), bias.float(), padding=1)
    got = fused.float().cpu().reshape(B, h, w, cout).permute(0, 3, 1, 2)
    check(got, ref, dtype, f"GroupNorm+SiLU -> conv {case}", scale=1.5)


def test_conv_gn_prologue_rejected_outside_the_slab_kernel():
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import pack_conv3x3
    dev = _dev()
    x = torch.randn(2 * 64, 128, device=dev).to(torch.bfloat16)        # N = 128: not a slab problem
    wp = pack_conv3x3(torch.randn(128, 128, 3, 3).to(torch.bfloat16)).to(dev)
    coef = torch.zeros(2, 2, 128, device=dev)
    with pytest.raises(RuntimeError, match="slab conv kernel"):
        ops.conv3x3(x, wp, 2, 8, 8, 128, a_coef=coef, a_silu=True)


LN_CASES = [  # (rows, C, row mean offset, row scale): the offset / scale stress the mean * u cancellation and E[x^2] - mean^2
    (4096, 320, 0.0, 1.0), (2100, 640, 0.5, 2.0), (1024, 1280, 3.0, 0.5), (384, 320, -8.0, 0.25),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", LN_CASES)
def test_gemm_layernorm_folded(dtype, case):
    """tg_gemm ln_u / ln_v (kernel_kind 6): LayerNorm folded into the projection that consumes it vs the fp32 reference
    ``F.linear(F.layer_norm(x), W, b)`` — the three shapes the UNet uses: attn2.to_q (N = C), attn1 q|k|v with the V^T split
    (N = 3C), FeedForward GEGLU (N = 8C, packed a|gate rows) — and vs the two-launch path (tg_layernorm then tg_gemm)."""
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import pack_geglu, pack_ln_linear
    dev = _dev()
    M, C, off, sc = case
    g = torch.Generator().manual_seed(M + C)
    x = ((torch.randn(M, C, generator=g) + off * torch.randn(M, 1, generator=g)) * sc).to(dtype)
    gamma, beta = (1 + 0.3 * torch.randn(C, generator=g)).to(dtype), (0.3 * torch.randn(C, generator=g)).to(dtype)
    eps = 1e-5
    xn = F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), eps)
    xd = x.to(dev)
    # (a) N = C, no bias
    w = rnd((C, C), dtype, g, 1 / math.sqrt(C))
    wl, u, v = pack_ln_linear(w.to(dev), None, gamma.to(dev), beta.to(dev))
    tm, tn, sp, kind = ops.gemm(xd, wl, M, C, C, ln=(u, v, eps), plan_only=True)
    assert kind == 6 and sp == 1
    got = ops.gemm(xd, wl, M, C, C, ln=(u, v, eps))
    check(got, xn @ w.float().t(), dtype, f"ln-folded to_q {case}", scale=1.5)
    two = ops.linear(ops.layernorm(xd, gamma.to(dev), beta.to(dev), eps), w.to(dev))
    check(got, two.float(), dtype, f"ln-folded vs two-launch {case}", scale=2.0)
    # the same fold with the row statistics precomputed (tg_layernorm_stats -> ln_rows)
    rows_st = ops.layernorm_stats(xd, eps)
    mu, var = x.float().mean(1), x.float().var(1, unbiased=False)
    rstd = (var + eps).rsqrt()
    assert torch.allclose(rows_st[:, 0].cpu(), rstd, rtol=2e-5, atol=0) and torch.allclose(rows_st[:, 1].cpu(), -rstd * mu, rtol=2e-5, atol=1e-6)
    got2 = ops.gemm(xd, wl, M, C, C, ln=(u, v, eps, rows_st))
    check(got2, xn @ w.float().t(), dtype, f"ln-folded (precomputed statistics) to_q {case}", scale=1.5)
    # (b) q | k | v^T split (rows_per_batch divides M)
    B = 2 if M % 2 == 0 else 1
    rows = M // B
    ldt = (rows + 7) // 8 * 8
    w3 = rnd((3 * C, C), dtype, g, 1 / math.sqrt(C))
    wl3, u3, v3 = pack_ln_linear(w3.to(dev), None, gamma.to(dev), beta.to(dev))
    qk = torch.zeros((M, 2 * C), dtype=dtype, device=dev)
    vt = torch.zeros((B, C, ldt), dtype=dtype, device=dev)
    ops.gemm(xd, wl3, M, 3 * C, C, rows_per_batch=rows, out=qk, n_split=2 * C, out_t=vt, ldt=ldt, ln=(u3, v3, eps))
    ref3 = xn @ w3.float().t()
    check(qk, ref3[:, :2 * C], dtype, f"ln-folded q|k {case}", scale=1.5)
    check(vt[:, :, :rows], ref3[:, 2 * C:].reshape(B, rows, C).permute(0, 2, 1), dtype, f"ln-folded v^T {case}", scale=1.5)
    # (c) GEGLU with bias
    N = 8 * C
    wf, bf = rnd((N, C), dtype, g, 1 / math.sqrt(C)), rnd((N,), dtype, g)
    y = xn @ wf.float().t() + bf.float()
    wp, bp = pack_geglu(wf.to(dev), bf.to(dev))
    wlg, ug, vg = pack_ln_linear(wp, bp, gamma.to(dev), beta.to(dev))
    gg = ops.gemm(xd, wlg, M, N, C, geglu=True, ln=(ug, vg, eps))
    assert gg.shape == (M, N // 2)
    check(gg, y[:, :N // 2] * F.gelu(y[:, N // 2:]), dtype, f"ln-folded geglu {case}", scale=1.5)
    gg2 = ops.gemm(xd, wlg, M, N, C, geglu=True, ln=(ug, vg, eps, rows_st))
    check(gg2, y[:, :N // 2] * F.gelu(y[:, N // 2:]), dtype, f"ln-folded geglu (precomputed statistics) {case}", scale=1.5)
    qk2 = torch.zeros((M, 2 * C), dtype=dtype, device=dev)
    vt2 = torch.zeros((B, C, ldt), dtype=dtype, device=dev)
    ops.gemm(xd, wl3, M, 3 * C, C, rows_per_batch=rows, out=qk2, n_split=2 * C, out_t=vt2, ldt=ldt, ln=(u3, v3, eps, rows_st))
    check(qk2, ref3[:, :2 * C], dtype, f"ln-folded q|k (precomputed statistics) {case}", scale=1.5)
    check(vt2[:, :, :rows], ref3[:, 2 * C:].reshape(B, rows, C).permute(0, 2, 1), dtype, f"ln-folded v^T (precomputed statistics) {case}", scale=1.5)


def test_gemm_layernorm_fold_argument_errors():
    from theatergen_amd import ops
    dev = _dev()
    x = torch.zeros(256, 64, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(64, 64, dtype=torch.bfloat16, device=dev)
    u = torch.zeros(64, dtype=torch.float32, device=dev)
    with pytest.raises(RuntimeError):
        ops.gemm(x, w, 256, 64, 64, ln=(u, u, 1e-5), res=x)            # no residual with the fold
    with pytest.raises(RuntimeError):
        ops.gemm(x, w, 256, 64, 64, ln=(u, u, 0.0))                    # eps must be positive
    with pytest.raises(RuntimeError):
        ops.gemm(x, w, 256, 64, 64, ln=(u, u, 1e-5), force_split_k=2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 50, 37), (1, 64, 320), (3, 128, 64), (2, 384, 320), (1, 2304, 2304)])
def test_transpose_bit_exact(dtype, shape):
    """tg_transpose: the 32 x 32 tile kernel (any shape) and the 128 x 64 tile kernel (rows % 128 == 0, cols % 64 == 0: the long
    self-attention maps of the reverse pass) move bits only."""
    from theatergen_amd import ops
    dev = _dev()
    b, r, c = shape
    g = torch.Generator().manual_seed(b * r + c)
    x = torch.randn((b, r, c), generator=g).to(dtype).to(dev)
    got = ops.transpose(x, b, r, c)
    assert got.shape == (b, c, r) and torch.equal(got, x.transpose(1, 2).contiguous())

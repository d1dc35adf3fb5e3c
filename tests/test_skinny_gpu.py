"""GPU: ``tg_skinny_gemm`` (csrc/tg_skinny.hip) through the C ABI, and the Resampler on it.

  * the kernel against the fp32 reference of the op: plain / bias / GELU / residual / LayerNorm-folded, ragged row counts, more than one row block, every
    chunk width (K / 64 in {1, 5, 12, 16, 20} and multi-chunk K), output routing into three segments (plain, plain with a per-batch row offset, transposed);
  * the Perceiver ``Resampler`` (reference ip_adapter/resampler.py:81-147) with the latent path on the skinny kernel vs the generic launch-per-op path of
    ``theatergen_amd.resampler`` on the same weights (the reference goldens of tests/test_hotpath_gpu.py run through the skinny path for the Plus configurations).
"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DTYPES = [torch.bfloat16, torch.float16]


def tols(dtype):
    return (3e-3, 1e-2) if dtype == torch.bfloat16 else (4e-4, 2.5e-3)


def check(got, ref, what, l2, mx):
    from tests import parity_metrics as pm
    return pm.check(got.float().cpu(), ref.float().cpu(), what, l2, mx)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,bias,act,res,ln", [
    (32, 1280, 1280, False, False, True, False),      # to_out + residual
    (32, 5120, 1280, False, True, False, True),       # FeedForward: norm + Linear + GELU
    (32, 1280, 5120, False, False, True, False),      # FeedForward out (multi-chunk K) + residual
    (32, 2048, 1280, True, False, False, False),      # proj_out + bias
    (16, 768, 768, True, True, True, True),           # SD-1.5 Plus width, b = 1, everything on
    (50, 96, 320, True, False, False, True),          # two row blocks, ragged; K / 64 = 5
    (7, 64, 64, False, False, False, True),           # K / 64 = 1
    (32, 1024, 1024, False, False, False, True),      # K / 64 = 16
    (3, 32, 192, True, False, True, False),           # 4 waves, chunk width 3
    (9, 64, 448, False, False, False, False),         # 4 waves, chunk width 1 (K / 64 = 7)
    (32, 640, 2560, False, True, False, False),       # 8 waves, K / 128 = 20
])
def test_skinny_gemm_vs_fp32(dtype, M, N, K, bias, act, res, ln):
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import pack_ln_linear, skinny_pack
    g = torch.Generator().manual_seed(M * 7 + N + K)
    x = (torch.randn(M, K, generator=g) * 1.3 + 0.4).to(dtype).to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype).to(DEV)
    b = torch.randn(N, generator=g).to(dtype).to(DEV) if bias else None
    r = torch.randn(M, N, generator=g).to(dtype).to(DEV) if res else None
    gamma = (1 + 0.2 * torch.randn(K, generator=g)).to(dtype).to(DEV)
    beta = (0.3 * torch.randn(K, generator=g)).to(dtype).to(DEV)
    xf = x.float()
    if ln:
        wp, u, v = pack_ln_linear(W, None, gamma, beta)
        got = ops.skinny_gemm(x, skinny_pack(wp), N, ln=(u, v, 1e-5), bias=b, act=ops.ACT_GELU if act else ops.ACT_NONE, res=r)
        xf = F.layer_norm(xf, (K,), gamma.float(), beta.float(), 1e-5)
    else:
        got = ops.skinny_gemm(x, skinny_pack(W), N, bias=b, act=ops.ACT_GELU if act else ops.ACT_NONE, res=r)
    ref = xf @ W.float().t()
    if b is not None:
        ref = ref + b.float()
    if r is not None:
        ref = ref + r.float()
    if act:
        ref = F.gelu(ref)
    l2, mx = tols(dtype)
    check(got, ref, f"skinny_gemm {M}x{N}x{K} bias={bias} act={act} res={res} ln={ln} {dtype}", l2, mx)


@pytest.mark.parametrize("dtype", DTYPES)
def test_skinny_gemm_routing(dtype):
    """[q | k | v] of 2 x 16 latents in one launch: q plain, k behind 257 image rows of each batch item's K block, v transposed behind 257 columns of V^T"""
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import skinny_pack
    g = torch.Generator().manual_seed(5)
    b, n2, n1, D, inner = 2, 16, 257, 768, 128
    L, ldt = n1 + n2, 280
    x = torch.randn(b * n2, D, generator=g).to(dtype).to(DEV)
    W = (torch.randn(3 * inner, D, generator=g) / D ** 0.5).to(dtype).to(DEV)
    q = torch.zeros(b * n2, inner, dtype=dtype, device=DEV)
    k = torch.full((b * L, inner), 7.0, dtype=dtype, device=DEV)
    vt = torch.full((b, inner, ldt), 9.0, dtype=dtype, device=DEV)
    es = x.element_size()
    ops.skinny_gemm(x, skinny_pack(W), 3 * inner, rows_per_batch=n2,
                    segs=[(q.data_ptr(), inner, n2 * inner, inner, 0), (k.data_ptr() + n1 * inner * es, inner, L * inner, 2 * inner, 0),
                          (vt.data_ptr() + n1 * es, ldt, inner * ldt, 3 * inner, 1)])
    ref = (x.float() @ W.float().t()).reshape(b, n2, 3 * inner)
    l2, mx = tols(dtype)
    check(q.reshape(b, n2, inner), ref[..., :inner], f"skinny routing q {dtype}", l2, mx)
    kk = k.reshape(b, L, inner)
    check(kk[:, n1:], ref[..., inner:2 * inner], f"skinny routing k {dtype}", l2, mx)
    check(vt[:, :, n1:L].transpose(1, 2), ref[..., 2 * inner:], f"skinny routing v^T {dtype}", l2, mx)
    assert (kk[:, :n1] == 7.0).all() and (vt[:, :, :n1] == 9.0).all() and (vt[:, :, L:] == 9.0).all(), "wrote outside its rows / columns"


def test_skinny_gemm_argument_errors():
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import skinny_pack
    x = torch.randn(8, 128, device=DEV).to(torch.bfloat16)
    W = torch.randn(64, 128, device=DEV).to(torch.bfloat16)
    with pytest.raises(RuntimeError, match="segments cover"):
        ops.skinny_gemm(x, skinny_pack(W), 64, rows_per_batch=8, segs=[(x.data_ptr(), 64, 0, 32, 0)])
    u = torch.zeros(64, device=DEV)
    x7 = torch.randn(8, 448, device=DEV).to(torch.bfloat16)          # K / 64 = 7: no single-chunk instance
    W7 = torch.randn(64, 448, device=DEV).to(torch.bfloat16)
    with pytest.raises(RuntimeError, match="keeps the whole row in registers"):
        ops.skinny_gemm(x7, skinny_pack(W7), 64, ln=(u, u, 1e-5))
    with pytest.raises(RuntimeError, match="multiple of 32"):
        ops.skinny_gemm(x, skinny_pack(W)[:48 * 128], 48)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name,kw,b", [
    ("sd15_plus", dict(dim=768, depth=4, dim_head=64, heads=12, num_queries=16, embedding_dim=1280, output_dim=768, ff_mult=4), 2),
    ("sdxl_plus", dict(dim=1280, depth=4, dim_head=64, heads=20, num_queries=16, embedding_dim=1280, output_dim=2048, ff_mult=4), 2),
    ("sdxl_plus_b3", dict(dim=1280, depth=2, dim_head=64, heads=20, num_queries=16, embedding_dim=1280, output_dim=2048, ff_mult=4), 3),
])
def test_resampler_skinny_vs_generic_path(dtype, name, kw, b):
    from theatergen_amd import weights as W
    from theatergen_amd.resampler import Resampler
    rs = Resampler(**kw)
    rs.load_state_dict(W.random_resampler_state_dict(seed=77, **kw))
    rs = rs.to(DEV, dtype)
    x = torch.randn(b, 257, kw["embedding_dim"], generator=torch.Generator().manual_seed(3)).to(DEV, dtype)
    x[-1].zero_()                                          # the zero-image item of the reference's uncond branch (ip_adapter.py:339-341)
    assert rs._skinny_ok(b, 257, kw["dim"])
    with torch.no_grad():
        fast = rs(x)
        os.environ["TG_RESAMPLER_SKINNY"] = "0"
        try:
            slow = rs(x)
        finally:
            del os.environ["TG_RESAMPLER_SKINNY"]
    l2, mx = tols(dtype)
    # two storage-dtype implementations with different rounding points (LayerNorm folded vs materialised, q | k | v from one accumulation) over 4 layers, each within
    # the golden tolerance of the fp32 reference (tests/test_hotpath_gpu.py::test_resampler_vs_reference_golden runs the Plus cases through the skinny path):
    # three single-op tolerances between them
    check(fast, slow, f"resampler skinny vs generic {name} {dtype}", 3 * l2, 3 * mx)

"""GPU: the row-chain kernels (csrc/tg_rowchain.hip) through the C ABI.

  * ``tg_rc_linear``: K = 320 projections with the token rows in registers (plain / + residual / LayerNorm-folded, bf16 and fp16, ragged
    row counts) vs the fp32 reference of the op and vs ``tg_gemm`` on the same inputs;
  * ``tg_rc_xattn`` + ``tg_rc_kv_pack``: norm2 + (decoupled text + image) cross-attention + to_out + residual of a first-level
    ``BasicTransformerBlock`` (reference models/attention.py:206-224, ip_adapter/attention_processor.py:445-529, 282-393) in one launch vs
    an fp32 restatement of those lines, for 0 / 4 / 16 image tokens and several IP scales;
  * a whole SD-1.5 first-level ``BasicTransformerBlock`` with the row-chain path ON vs OFF (the three-launch path the goldens pin).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DTYPES = [torch.bfloat16, torch.float16]


def tols(dtype):
    # (rel-L2, max|err| / max|ref|): one storage rounding of the output + the rounding of the intermediate q / P / O tensors
    return (3e-3, 1e-2) if dtype == torch.bfloat16 else (4e-4, 2.5e-3)


def check(got, ref, what, l2, mx):
    from tests import parity_metrics as pm
    return pm.check(got.float().cpu(), ref.float().cpu(), what, l2, mx)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,res,ln", [(8192, 320, True, False), (8192, 320, False, True), (4096, 960, False, True),
                                        (1000, 320, True, True), (33, 64, False, False), (8192, 2560, False, False)])
def test_rc_linear_vs_fp32_and_tg_gemm(dtype, M, N, res, ln):
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import pack_ln_linear, rc_pack
    g = torch.Generator().manual_seed(M + N)
    K = 320
    x = (torch.randn(M, K, generator=g) * 1.5 + 0.3).to(dtype).to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype).to(DEV)
    bias = torch.randn(N, generator=g).to(dtype).to(DEV)
    gamma = (1 + 0.2 * torch.randn(K, generator=g)).to(dtype).to(DEV)
    beta = (0.1 * torch.randn(K, generator=g)).to(dtype).to(DEV)
    r = torch.randn(M, N, generator=g).to(dtype).to(DEV) if res else None
    if ln:
        Wp, u, v = pack_ln_linear(W, bias, gamma, beta)
        wpk = rc_pack(Wp, v, u)
        ref = F.linear(F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-5), W.float(), bias.float())
        old = ops.linear(x, Wp, None, ln=(u, v, 1e-5)) if not res else None      # tg_gemm's fold takes no residual
    else:
        wpk = rc_pack(W, bias.float())
        ref = F.linear(x.float(), W.float(), bias.float())
        old = ops.linear(x, W, bias, res=r)
    if res:
        ref = ref + r.float()
    l2, mx = tols(dtype)
    for variant in (0, 1, 3):
        got = ops.rc_linear(x, wpk, N, res=r, ln_eps=1e-5 if ln else None, variant=variant)
        check(got, ref, f"rc_linear {M}x{N} res={res} ln={ln} v{variant} {dtype}", l2 * (1.5 if ln else 1), mx * (1.5 if ln else 1))
    # the two kernels round the same fp32 sums (different accumulation order): they agree far inside the tolerance to fp32
    if old is not None:
        check(got, old, f"rc_linear vs tg_gemm {M}x{N} {dtype}", l2, mx)


def test_rc_linear_rejects_other_shapes():
    from theatergen_amd import ops
    x = torch.zeros(64, 1280, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError):
        ops.rc_linear(x, torch.zeros((64 // 32) * 64 * 1280, dtype=torch.uint8, device=DEV), 64)        # K = 1280: a row does not fit the registers
    x6 = torch.zeros(64, 640, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError):
        ops.rc_linear(x6, torch.zeros((128 // 32) * 64 * 640, dtype=torch.uint8, device=DEV), 128)      # K = 640: round 4's variant was removed


def _xattn_case(B, N, T, dtype, seed, ip_w=0.4):
    C, H, D, L = 320, 8, 40, 77
    g = torch.Generator().manual_seed(seed)
    M = B * N
    t = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    h = (t(M, C, sc=1.2) + 0.2).to(dtype).to(DEV)
    wq = t(C, C, sc=C ** -0.5).to(dtype).to(DEV)
    wo = t(C, C, sc=C ** -0.5).to(dtype).to(DEV)
    bo = t(C, sc=0.1).to(dtype).to(DEV)
    gamma = (1 + 0.2 * t(C)).to(dtype).to(DEV)
    beta = t(C, sc=0.1).to(dtype).to(DEV)
    k = t(B * L, C).to(dtype).to(DEV)
    v = t(B, L, C).to(dtype).to(DEV)
    kip = t(B * max(T, 1), C).to(dtype).to(DEV)
    vip = t(B, max(T, 1), C).to(dtype).to(DEV)
    ldt, ldi = 80, 8 * ((max(T, 1) + 7) // 8)
    vt = torch.zeros(B, C, ldt, device=DEV, dtype=dtype); vt[:, :, :L] = v.transpose(1, 2)
    vtip = torch.zeros(B, C, ldi, device=DEV, dtype=dtype); vtip[:, :, :max(T, 1)] = vip.transpose(1, 2)
    scale = D ** -0.5
    x = F.layer_norm(h.float(), (C,), gamma.float(), beta.float(), 1e-5)
    q = (x @ wq.float().T).reshape(B, N, H, D).permute(0, 2, 1, 3)
    kk = k.float().reshape(B, L, H, D).permute(0, 2, 1, 3)
    vv = v.float().reshape(B, L, H, D).permute(0, 2, 1, 3)
    o = torch.softmax(q @ kk.transpose(-1, -2) * scale, -1) @ vv
    if T:
        ki = kip.float().reshape(B, T, H, D).permute(0, 2, 1, 3)
        vi = vip.float().reshape(B, T, H, D).permute(0, 2, 1, 3)
        o = o + ip_w * torch.softmax(q @ ki.transpose(-1, -2) * scale, -1) @ vi
    ref = o.permute(0, 2, 1, 3).reshape(M, C) @ wo.float().T + bo.float() + h.float()
    return dict(h=h, wq=wq, wo=wo, bo=bo, gamma=gamma, beta=beta, k=k, vt=vt, ldt=ldt, kip=kip, vtip=vtip, ldi=ldi, L=L, scale=scale, ref=ref)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N,T,ip_w", [(2, 256, 4, 0.4), (2, 256, 0, 0.0), (1, 512, 16, 1.0), (3, 128, 4, 0.0), (2, 4096, 4, 0.1)])
def test_rc_xattn_vs_fp32_reference(dtype, B, N, T, ip_w):
    from theatergen_amd import ops, rowchain
    c = _xattn_case(B, N, T, dtype, seed=7 * B + N + T, ip_w=ip_w)
    wqp = rowchain.pack_xattn_q(c["wq"], None, c["gamma"], c["beta"], c["scale"])
    wop = rowchain.pack_xattn_out(c["wo"], c["bo"])
    kv = ops.rc_kv_pack(c["k"], c["vt"], c["ldt"], c["L"], c["kip"] if T else None, c["vtip"] if T else None, c["ldi"], T, B)
    w = torch.full((1,), ip_w, device=DEV)
    got = ops.rc_xattn(c["h"], wqp, kv, wop, N, 1e-5, T, ip_scale=w if T else None)
    l2, mx = tols(dtype)
    check(got, c["ref"], f"rc_xattn B{B} N{N} T{T} w{ip_w} {dtype}", l2, mx)
    if T:
        # the IP scale is read from the device at run time: the same launch with another value (graph-replay contract)
        w.fill_(0.0)
        got0 = ops.rc_xattn(c["h"], wqp, kv, wop, N, 1e-5, T, ip_scale=w)
        c0 = _xattn_case(B, N, T, dtype, seed=7 * B + N + T, ip_w=0.0)
        check(got0, c0["ref"], f"rc_xattn scale 0 B{B} N{N} T{T} {dtype}", l2, mx)


def test_rc_xattn_argument_errors():
    from theatergen_amd import ops
    z = torch.zeros(256, 320, dtype=torch.bfloat16, device=DEV)
    u8 = torch.zeros(10 * 21 * 1024 + 3072, dtype=torch.uint8, device=DEV)
    kv = torch.zeros(1, 8, 24 * 512, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError):
        ops.rc_xattn(z, u8, kv, u8, 256, 1e-5, 5)            # 5 image tokens: not built
    with pytest.raises(RuntimeError):
        ops.rc_xattn(z, u8, kv, u8, 100, 1e-5, 4)            # a workgroup's 128 rows must share a key set
    with pytest.raises(RuntimeError):
        ops.rc_xattn(z, u8, kv, u8, 256, 1e-5, 4, text_len=64)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ip", [True, False])
def test_first_level_block_row_chain_on_vs_off(dtype, ip, monkeypatch):
    """A whole SD-1.5 first-level BasicTransformerBlock (320 channels, 8 heads, 77 + 4 tokens): the row-chain path (rc_linear for
    to_out, tg_rc_xattn for norm2 + attn2 + residual) against the three-launch path on the same weights and inputs."""
    from theatergen_amd import rowchain
    from theatergen_amd.attention_processor import AttnProcessor, IPAttnProcessor
    from theatergen_amd.unet import BasicTransformerBlock
    torch.manual_seed(3)
    C, B, N, ctx = 320, 2, 4096, 768
    blk = BasicTransformerBlock(C, 8, 40, ctx)
    for prm in blk.parameters():
        if prm.ndim == 1:
            prm.data.normal_(0, 0.1)
    blk.norm1.weight.data.add_(1.0); blk.norm2.weight.data.add_(1.0); blk.norm3.weight.data.add_(1.0)
    if ip:
        proc = IPAttnProcessor(C, ctx, scale=0.4, num_tokens=4)
        blk.attn2.set_processor(proc)
    blk = blk.to(DEV, dtype)
    x = (torch.randn(B * N, C) * 1.1).to(DEV, dtype)
    enc = (torch.randn(B, 81 if ip else 77, ctx) * 0.5).to(DEV, dtype)
    monkeypatch.setattr(rowchain, "ENABLED", False)
    ref = blk.run(x, B, N, enc, {})
    monkeypatch.setattr(rowchain, "ENABLED", True)
    monkeypatch.setattr(rowchain, "MIN_ROWS", 1024)
    monkeypatch.setattr(rowchain, "MIN_ROWS_CHAIN", 1024)
    got = blk.run(x, B, N, enc, {})
    l2, mx = tols(dtype)
    check(got, ref, f"first-level block row-chain on/off ip={ip} {dtype}", 1.5 * l2, 1.5 * mx)
    if ip:
        proc.scale = 0.0                      # IPAdapter.set_scale between characters: device scalar, no repack
        monkeypatch.setattr(rowchain, "ENABLED", False)
        ref0 = blk.run(x, B, N, enc, {})
        monkeypatch.setattr(rowchain, "ENABLED", True)
        got0 = blk.run(x, B, N, enc, {})
        check(got0, ref0, f"first-level block row-chain scale 0 {dtype}", 1.5 * l2, 1.5 * mx)
        assert (ref0.float() - ref.float()).abs().max() > 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,inner,proj", [(256, 128, False), (1000, 256, True), (8192, 1280, True), (4096, 1280, False)])
def test_rc_ff_vs_fp32_reference(dtype, M, inner, proj):
    """norm3 + GEGLU feed-forward + residual (+ proj_out + residual) in one launch vs the fp32 restatement of models/attention.py:226-236,
    328-338 (erf GELU) and models/transformer_2d.py:316-327"""
    from theatergen_amd import ops, rowchain
    from theatergen_amd.weights_pack import rc_pack_tiles
    g = torch.Generator().manual_seed(M + inner)
    C = 320
    t = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    h = (t(M, C, sc=1.2) + 0.2).to(dtype).to(DEV)
    x0 = t(M, C).to(dtype).to(DEV)
    w1 = t(2 * inner, C, sc=C ** -0.5).to(dtype).to(DEV); b1 = t(2 * inner, sc=0.2).to(dtype).to(DEV)
    w2 = t(C, inner, sc=inner ** -0.5).to(dtype).to(DEV); b2 = t(C, sc=0.1).to(dtype).to(DEV)
    wp = t(C, C, sc=C ** -0.5).to(dtype).to(DEV); bp = t(C, sc=0.1).to(dtype).to(DEV)
    gamma = (1 + 0.2 * t(C)).to(dtype).to(DEV); beta = t(C, sc=0.1).to(dtype).to(DEV)
    x = F.layer_norm(h.float(), (C,), gamma.float(), beta.float(), 1e-5)
    pr = x @ w1.float().T + b1.float()
    h3 = (pr[:, :inner] * F.gelu(pr[:, inner:])) @ w2.float().T + b2.float() + h.float()
    ref = h3 @ wp.float().T + bp.float() + x0.float() if proj else h3
    s1, s2, bb2 = rowchain.pack_ff(w1, b1, gamma, beta, w2, b2)
    got = ops.rc_ff(h, s1, s2, bb2, inner, 1e-5, wpo=rc_pack_tiles(wp, bp.float()) if proj else None, res0=x0 if proj else None)
    l2, mx = tols(dtype)
    # three storage roundings on the way (normalised rows, hidden activations, h3 before proj_out): 1.5x the single-op tolerance
    check(got, ref, f"rc_ff M{M} inner{inner} proj{proj} {dtype}", 1.5 * l2, 1.5 * mx)


@pytest.mark.parametrize("dtype", DTYPES)
def test_first_level_transformer_row_chain_on_vs_off(dtype, monkeypatch):
    """Transformer2DModel of SD-1.5's first level (GroupNorm -> proj_in -> block -> proj_out + residual) with every row-chain launch ON
    (rc_linear, rc_xattn, rc_ff incl. proj_out) vs OFF on the same weights / inputs; then modes one at a time."""
    from theatergen_amd import rowchain
    from theatergen_amd.attention_processor import IPAttnProcessor
    from theatergen_amd.unet import Transformer2DModel, _Act
    torch.manual_seed(5)
    C, B, Hh, ctx = 320, 2, 64, 768
    tf = Transformer2DModel(8, 40, C, 1, ctx, 32, False)
    for prm in tf.parameters():
        if prm.ndim == 1:
            prm.data.normal_(0, 0.1)
    for nrm in (tf.norm, tf.transformer_blocks[0].norm1, tf.transformer_blocks[0].norm2, tf.transformer_blocks[0].norm3):
        nrm.weight.data.add_(1.0)
    tf.transformer_blocks[0].attn2.set_processor(IPAttnProcessor(C, ctx, scale=0.4, num_tokens=4))
    tf = tf.to(DEV, dtype)
    x = _Act((torch.randn(B * Hh * Hh, C) * 1.1).to(DEV, dtype), B, Hh, Hh, C)
    enc = (torch.randn(B, 81, ctx) * 0.5).to(DEV, dtype)
    monkeypatch.setattr(rowchain, "ENABLED", False)
    ref = tf.run(x, enc, {}).t
    monkeypatch.setattr(rowchain, "ENABLED", True)
    monkeypatch.setattr(rowchain, "MIN_ROWS", 1024)
    monkeypatch.setattr(rowchain, "MIN_ROWS_CHAIN", 1024)
    l2, mx = tols(dtype)
    for mode in (15, 1, 2, 4, 8):
        monkeypatch.setattr(rowchain, "MODE", mode)
        got = tf.run(x, enc, {}).t
        check(got, ref, f"first-level Transformer2DModel row-chain mode {mode} {dtype}", 2 * l2, 2 * mx)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N", [(2, 256), (3, 128), (2, 4096)])
def test_rc_front_vs_fp32_reference(dtype, B, N):
    """GroupNorm (coefficients) + proj_in + LayerNorm1 + q | k | v in one launch vs the fp32 restatement of models/transformer_2d.py:285-296
    and models/attention.py:186-204; Q | K token-major, V transposed per batch item (what tg_attention reads)"""
    from theatergen_amd import ops
    from theatergen_amd.weights_pack import pack_ln_linear, rc_pack_tiles
    g = torch.Generator().manual_seed(B * N)
    C, M = 320, B * N
    t = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    x = (t(M, C, sc=1.3) + 0.4).to(dtype).to(DEV)
    gg = (1 + 0.2 * t(C)).to(dtype).to(DEV); gb = t(C, sc=0.1).to(dtype).to(DEV)
    win = t(C, C, sc=C ** -0.5).to(dtype).to(DEV); bin_ = t(C, sc=0.1).to(dtype).to(DEV)
    wq = t(3 * C, C, sc=C ** -0.5).to(dtype).to(DEV)
    lg = (1 + 0.2 * t(C)).to(dtype).to(DEV); lb = t(C, sc=0.1).to(dtype).to(DEV)
    xg = F.group_norm(x.float().reshape(B, N, C).permute(0, 2, 1), 32, gg.float(), gb.float(), 1e-6).permute(0, 2, 1).reshape(M, C)
    y = xg @ win.float().T + bin_.float()
    qkv = F.layer_norm(y, (C,), lg.float(), lb.float(), 1e-5) @ wq.float().T
    coef = ops.groupnorm_coef(x, B, N, 32, 1e-6, gg, gb)
    Wp, u, v = pack_ln_linear(wq, None, lg, lb)
    gy, gqk, gvt, ldt = ops.rc_front(x, coef, rc_pack_tiles(win, bin_.float()), rc_pack_tiles(Wp, v, u), N, 1e-5)
    l2, mx = tols(dtype)
    check(gy, y, f"rc_front y B{B} N{N} {dtype}", 1.5 * l2, 1.5 * mx)
    check(gqk, qkv[:, :640], f"rc_front qk B{B} N{N} {dtype}", 2 * l2, 2 * mx)
    check(gvt[:, :, :N].permute(0, 2, 1).reshape(M, C), qkv[:, 640:], f"rc_front v^T B{B} N{N} {dtype}", 2 * l2, 2 * mx)

"""Tensor-level wrappers over the C ABI (one function per kernel family).

PyTorch is used ONLY for device memory (``torch.empty`` / views / data_ptr) and the current HIP stream;
every computation is a call into libtheatergen_hip.so.  Activations are token-major 2-D views
``[rows, channels]`` (``rows = batch * h * w``) in bf16 or fp16.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_QUICK_GELU, ACT_SILU, AttnDesc, GemmDesc  # noqa: F401

_workspaces = {}


def _dt(t):
    if t.dtype == torch.bfloat16:
        return _lib.TG_BF16
    if t.dtype == torch.float16:
        return _lib.TG_F16
    raise RuntimeError(f"theatergen_amd: unsupported activation dtype {t.dtype} (bf16 / fp16 only)")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def _need_cuda(t):
    if not t.is_cuda:
        raise RuntimeError("theatergen_amd: tensors must live on the GPU (no CPU fallback)")


_ws_slot = 0


class workspace_slot:
    """Kernels launched inside this context take their scratch (GroupNorm partial sums, K-split partial tiles) from slot
    ``slot``: denoising chains that run CONCURRENTLY on different streams must not share one scratch buffer."""

    def __init__(self, slot):
        self.slot = int(slot)

    def __enter__(self):
        global _ws_slot
        self._prev, _ws_slot = _ws_slot, self.slot

    def __exit__(self, *exc):
        global _ws_slot
        _ws_slot = self._prev
        return False


def workspace(nbytes, device):
    """Persistent fp32 scratch per (device, slot) (split-K partials, GroupNorm partial sums)."""
    key = (device.index, "ws", _ws_slot)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty(max(int(nbytes) // 4 + 1, 1 << 20), dtype=torch.float32, device=device)
        _workspaces[key] = buf
    return buf


def gemm(a0, w, M, N, K, *, mode=0, a1=None, c0=None, c1=0, conv=None, bias=None, bvec=None, rows_per_batch=0,
         res=None, act=ACT_NONE, out_scale=1.0, out=None, n_split=0, out_t=None, ldt=0, force_split_k=0, force_tile=0,
         a_rows_per_batch=0, a_batch_stride=0, geglu=False, pad_mode=0, a_coef=None, a_silu=False, plan_only=False, ln=None, lda=0, ldw=0,
         gn_out=None):
    """out[M, N] = epilogue(A[M, K] @ W[N, K]^T); see tg_gemm in include/theatergen_hip.h.
    ``conv`` = (batch, in_h, in_w, out_h, out_w, stride, upsample) for mode 1."""
    _need_cuda(a0)
    L = _lib.lib()
    d = GemmDesc()
    d.dtype = _dt(a0)
    d.mode = mode
    d.a0, d.a1 = _ptr(a0), _ptr(a1)
    d.c0 = int(c0 if c0 is not None else (K if mode == 0 else K // 9))
    d.c1 = int(c1)
    if conv is not None:
        d.batch, d.in_h, d.in_w, d.out_h, d.out_w, d.stride, d.upsample = [int(v) for v in conv]
    d.w = _ptr(w)
    d.M, d.N, d.K = int(M), int(N), int(K)
    d.bias = _ptr(bias)
    d.bvec = _ptr(bvec)
    d.ldbvec = int(bvec.stride(0)) if bvec is not None else 0
    d.rows_per_batch = int(rows_per_batch)
    d.res = _ptr(res)
    d.ldres = int(res.stride(0)) if res is not None else 0
    d.act = act
    d.geglu = 1 if geglu else 0
    d.out_scale = float(out_scale)
    n_main = n_split if n_split > 0 else (N // 2 if geglu else N)
    if out is None:
        out = torch.empty((M, n_main), dtype=a0.dtype, device=a0.device)
    d.out = _ptr(out)
    d.ldc = int(out.stride(0))
    d.n_split = int(n_split)
    d.out_t = _ptr(out_t)
    d.ldt = int(ldt)
    d.force_split_k = force_split_k
    d.force_tile = force_tile
    d.a_rows_per_batch, d.a_batch_stride = int(a_rows_per_batch), int(a_batch_stride)
    d.pad_mode = int(pad_mode)
    d.a_coef = _ptr(a_coef)
    d.a_silu = 1 if a_silu else 0
    d.lda, d.ldw = int(lda), int(ldw)                    # row pitches of A / W when the caller padded them (plain single-source GEMM only; 0 = K)
    if ln is not None:                  # (u fp32 [N], v fp32 [N], eps[, rows]): LayerNorm of the A rows folded into this GEMM (``pack_ln_linear``);
        d.ln_u, d.ln_v, d.ln_eps = _ptr(ln[0]), _ptr(ln[1]), float(ln[2])     # rows = ``layernorm_stats`` output, None: taken inside the kernel
        if len(ln) > 3 and ln[3] is not None:
            if ln[3].dtype != torch.float32 or ln[3].numel() != 2 * M or not ln[3].is_contiguous():
                raise RuntimeError("gemm: ln rows must be the contiguous fp32 [M, 2] tensor of layernorm_stats")
            d.ln_rows = _ptr(ln[3])
    if gn_out is not None and not plan_only:
        # gn_out = {"groups": G}: GroupNorm(G) partial sums of the OUTPUT from the epilogue where the planner's kernel writes them (the two-wave slab conv,
        # unsplit); filled in with "partials" (fp32 [batch, nblk, G, 2]) and "nblk" for ``groupnorm_from_partials``; left alone otherwise.
        d.out_gn_groups = int(gn_out["groups"])
        nb = L.tg_gemm_gn_partial_blocks(C.byref(d))
        if nb > 0:
            gn_out["partials"] = torch.empty((int(conv[0]), nb, d.out_gn_groups, 2), dtype=torch.float32, device=a0.device)
            gn_out["nblk"] = nb
            d.out_gn_partials = gn_out["partials"].data_ptr()
        else:
            d.out_gn_groups = 0
    if plan_only:
        tm, tn, sp, kk = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(L.tg_gemm_plan(C.byref(d), C.byref(tm), C.byref(tn), C.byref(sp), C.byref(kk)))
        return tm.value, tn.value, sp.value, kk.value
    need = L.tg_gemm_workspace_bytes(C.byref(d))
    if need < 0:
        _lib.check(-1)
    if need > 0:
        ws = workspace(need, a0.device)
        d.workspace = ws.data_ptr()
        d.workspace_bytes = ws.numel() * 4
    if _gemm_profile is None:
        _lib.check(L.tg_gemm(C.byref(d), _stream()))
        return out
    # profiling mode (bench.py roofline leg): HIP events on the launch stream around this one launch
    tm, tn, sp, kk = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    _lib.check(L.tg_gemm_plan(C.byref(d), C.byref(tm), C.byref(tn), C.byref(sp), C.byref(kk)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(L.tg_gemm(C.byref(d), _stream()))
    e1.record()
    if kk.value == 2:
        kname = "conv_halo_kernel<128x128>"
    elif kk.value == 4:
        # round 6: tiles of one 64 / 32 / 16-wide patch — the whole row or a patch of a wider map — run on conv_slab_pp_kernel (csrc/tg_conv_slab.hip: TG_SLAB_PP, default 3)
        spp = os.environ.get("TG_SLAB_PP", "3")
        ow = int(conv[4]) if conv is not None else 0
        pw = ow if ow in (16, 32, 64) else (64 if ow % 64 == 0 else 32 if ow % 32 == 0 else 16 if ow % 16 == 0 else 8)
        two_wave = ((pw == 64 and spp != "0") or (pw in (16, 32) and spp not in ("0", "1"))) and (pw == ow or spp not in ("0", "1", "2"))
        kname = ("conv_slab_pp_kernel" if two_wave and int(conv[2]) == ow else "conv_slab_kernel") + f"<{tm.value}x{tn.value}>" + ("+gn" if a_coef is not None else "")
    elif kk.value == 3:
        kname = f"bt_gemm_kernel<{tm.value}x{tn.value}>"
    elif kk.value == 6:
        kname = f"gemm_glds_kernel<plain+ln,{tm.value}x{tn.value}>"
    elif kk.value == 7:
        kname = ("pp160_gemm_kernel" if tn.value == 160 else "pp_gemm_kernel") + f"<{tm.value}x{tn.value}" + (",geglu" if geglu else "") + (",ln" if ln is not None else "") + ">"
    else:
        kname = f"gemm_glds_kernel<{'conv' if mode == 1 else 'plain'},{tm.value}x{tn.value}>"
    _gemm_profile.append(dict(kernel=kname, splits=sp.value,
                              M=int(M), N=int(N), K=int(K), flops=2.0 * M * N * K, events=(e0, e1),
                              has_res=res is not None, n_out=int(n_main if n_split <= 0 else N)))
    return out


_gemm_profile = None


def _profiled(launch, kernel, M, N, K, flops, has_res=False, alg_bytes=None):
    """run ``launch()``; in profiling mode (bench.py roofline leg) bracket it with HIP events on the launch stream and record it"""
    if _gemm_profile is None:
        launch()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    _gemm_profile.append(dict(kernel=kernel, splits=1, M=int(M), N=int(N), K=int(K), flops=float(flops), events=(e0, e1),
                              has_res=bool(has_res), n_out=int(N), **({"alg_bytes": float(alg_bytes)} if alg_bytes is not None else {})))


def gemm_profile_start():
    global _gemm_profile
    _gemm_profile = []


def gemm_profile_stop():
    """-> list of per-launch records with ``ms`` filled in (call after torch.cuda.synchronize())."""
    global _gemm_profile
    recs, _gemm_profile = _gemm_profile, None
    for r in recs or []:
        e0, e1 = r.pop("events")
        r["ms"] = e0.elapsed_time(e1)
    return recs or []


def linear(x, w, bias=None, **kw):
    """x [rows, K] (row pitch = K) @ w[N, K]^T"""
    M, K = x.shape
    N = w.shape[0]
    if x.stride(0) != K or w.shape[1] != K or w.stride(0) != K:
        # padded row pitches (round 5: the FeedForward hidden tensor and its packed net.2 weight): views [:, :K] of wider buffers
        assert x.stride(1) == 1 and w.stride(1) == 1 and w.shape[1] == K and x.stride(0) >= K and w.stride(0) >= K
        return gemm(x, w, M, N, K, bias=bias, lda=x.stride(0), ldw=w.stride(0), **kw)
    return gemm(x, w, M, N, K, bias=bias, **kw)


def rc_linear(x, wpk, N, *, res=None, ln_eps=None, out=None, variant=0):
    """Row-chain projection (csrc/tg_rowchain.hip): out = [LayerNorm-folded] x @ W^T + v (+ res) with the token rows in registers;
    ``wpk`` = ``weights_pack.rc_pack(W, v, u)`` (uint8 chunk stream), x [M, 320]."""
    from ._lib import RcLinearDesc
    _need_cuda(x)
    M, K = x.shape
    assert x.stride(1) == 1 and wpk.dtype == torch.uint8
    if K != 320:
        raise RuntimeError(f"rc_linear: K = {K} (the row-chain projection keeps a 320-channel token row in registers)")
    assert wpk.numel() == (N // 64) * (128 * K + 1024), (wpk.numel(), N, K)
    if out is None:
        out = torch.empty(M, N, dtype=x.dtype, device=x.device)
    d = RcLinearDesc()
    d.dtype = _dt(x)
    d.x, d.ldx, d.wpk = _ptr(x), int(x.stride(0)), _ptr(wpk)
    d.res, d.ldres = (_ptr(res), int(res.stride(0))) if res is not None else (None, 0)
    d.out, d.ldc = _ptr(out), int(out.stride(0))
    d.M, d.N, d.K = int(M), int(N), int(K)
    d.ln, d.ln_eps = (1, float(ln_eps)) if ln_eps is not None else (0, 0.0)
    d.variant = int(variant)
    _profiled(lambda: _lib.check(_lib.lib().tg_rc_linear(C.byref(d), _stream())), "rc_linear_kernel<320>" + ("+ln" if ln_eps is not None else ""),
              M, N, K, 2.0 * M * N * K, res is not None)
    return out


def rc_kv_pack(k, vt, ldt, L, kip, vtip, ldi, T, batch, out=None):
    """text K [batch * L, 320] / V^T [batch, 320, ldt] (+ image K / V^T) -> the fragment blocks ``rc_xattn`` streams: uint8-free view
    [batch, 8, 24 * 512] of the storage dtype"""
    _need_cuda(k)
    if out is None:
        out = torch.empty((batch, 8, 24 * 512), dtype=k.dtype, device=k.device)
    _lib.check(_lib.lib().tg_rc_kv_pack(_dt(k), int(batch), _ptr(k), _ptr(vt), int(ldt), int(L), _ptr(kip), _ptr(vtip), int(ldi), int(T),
                                       _ptr(out), _stream()))
    return out


def xq_kv_pack(k, vt, ldt, L, kip, vtip, ldi, T, batch, C_, head_dim, out=None):
    """text K [batch * L, C] / V^T [batch, C, ldt] (+ image K / V^T) -> the MFMA fragments ``xq_attn`` streams (uint8 blob; tg_xq_kv_pack)"""
    _need_cuda(k)
    need = _lib.lib().tg_xq_kv_bytes(int(batch), int(C_), int(head_dim))
    if need < 0:
        raise RuntimeError(f"xq_kv_pack: C = {C_}, head_dim = {head_dim} not supported")
    if out is None:
        out = torch.empty(need, dtype=torch.uint8, device=k.device)
    assert out.numel() == need and out.dtype == torch.uint8
    _lib.check(_lib.lib().tg_xq_kv_pack(_dt(k), int(batch), int(C_), int(head_dim), _ptr(k), _ptr(vt), int(ldt), int(L), _ptr(kip), _ptr(vtip),
                                       int(ldi), int(T), _ptr(out), _stream()))
    return out


def xq_attn(x, wq, ln_u, ln_v, ln_eps, kv, head_dim, rows_per_batch, text_len, ip_tokens, ip_scale=None, out=None):
    """norm2 + attn2.to_q + (decoupled text + image) cross-attention of an inner-level block in one launch -> O [M, C]; see tg_xq_attn"""
    from ._lib import XqAttnDesc
    _need_cuda(x)
    M, Cc = x.shape
    assert x.stride(1) == 1 and x.stride(0) == Cc and wq.shape == (Cc, Cc)
    if out is None:
        out = torch.empty(M, Cc, dtype=x.dtype, device=x.device)
    d = XqAttnDesc()
    d.dtype = _dt(x)
    d.x, d.ldx, d.wq = _ptr(x), int(x.stride(0)), _ptr(wq)
    d.ln_u, d.ln_v, d.ln_eps = _ptr(ln_u), _ptr(ln_v), float(ln_eps)
    d.kv, d.ip_scale = _ptr(kv), _ptr(ip_scale)
    d.out, d.ldc = _ptr(out), int(out.stride(0))
    d.M, d.C, d.head_dim, d.rows_per_batch, d.text_len, d.ip_tokens = int(M), int(Cc), int(head_dim), int(rows_per_batch), int(text_len), int(ip_tokens)
    heads = Cc // head_dim
    L_ = text_len + ip_tokens
    flops = 2.0 * M * Cc * Cc + 4.0 * M * L_ * Cc
    _profiled(lambda: _lib.check(_lib.lib().tg_xq_attn(C.byref(d), _stream())), f"gemm_glds_kernel<plain+ln+xattn d{head_dim},128x160>", M, Cc, Cc, flops,
              alg_bytes=2.0 * (2 * M * Cc + Cc * Cc) + 2.0 * (M // rows_per_batch) * 2 * L_ * Cc)
    return out


def rc_xattn(h, wq, kv, wo, rows_per_batch, ln_eps, ip_tokens, ip_scale=None, out=None, text_len=77):
    """norm2 + cross-attention (+ decoupled image keys) + to_out + residual in one launch; see tg_rc_xattn"""
    from ._lib import RcXattnDesc
    _need_cuda(h)
    M, Cc = h.shape
    assert Cc == 320 and h.stride(1) == 1
    if out is None:
        out = torch.empty(M, 320, dtype=h.dtype, device=h.device)
    d = RcXattnDesc()
    d.dtype = _dt(h)
    d.h, d.ldh = _ptr(h), int(h.stride(0))
    d.wq, d.kv, d.wo = _ptr(wq), _ptr(kv), _ptr(wo)
    d.out, d.ldc = _ptr(out), int(out.stride(0))
    d.M, d.rows_per_batch = int(M), int(rows_per_batch)
    d.text_len, d.ip_tokens = int(text_len), int(ip_tokens)
    d.ln_eps = float(ln_eps)
    d.ip_scale = _ptr(ip_scale)
    # algorithmic work: to_q + to_out (2 x 2 M 320^2) + scores and PV over the real keys (4 M 320 (L + T))
    _profiled(lambda: _lib.check(_lib.lib().tg_rc_xattn(C.byref(d), _stream())), f"rc_xattn_kernel<77+{int(ip_tokens)}>",
              M, 320, 320, 4.0 * M * 320 * 320 + 4.0 * M * 320 * (int(text_len) + int(ip_tokens)), True,
              alg_bytes=2.0 * (3 * M * 320 + 2 * 320 * 320) + 2.0 * kv.numel())     # rows in, residual re-read, rows out; to_q, to_out; K / V^T fragments
    return out


def rc_ff(h, w1, w2, b2, inner, ln_eps, wpo=None, res0=None, out=None, dbg=0):
    """norm3 + GEGLU feed-forward + residual (+ proj_out + its residual) in one launch; see tg_rc_ff"""
    from ._lib import RcFfDesc
    _need_cuda(h)
    M, Cc = h.shape
    assert Cc == 320 and h.stride(1) == 1
    if out is None:
        out = torch.empty(M, 320, dtype=h.dtype, device=h.device)
    d = RcFfDesc()
    d.dtype = _dt(h)
    d.h, d.ldh = _ptr(h), int(h.stride(0))
    d.w1, d.w2, d.b2, d.wpo = _ptr(w1), _ptr(w2), _ptr(b2), _ptr(wpo)
    d.res0, d.ldres = (_ptr(res0), int(res0.stride(0))) if res0 is not None else (None, 0)
    d.out, d.ldc = _ptr(out), int(out.stride(0))
    d.M, d.inner, d.ln_eps, d.dbg = int(M), int(inner), float(ln_eps), int(dbg)
    fl = 2.0 * M * 320 * (3 * int(inner)) + (2.0 * M * 320 * 320 if wpo is not None else 0.0)
    _profiled(lambda: _lib.check(_lib.lib().tg_rc_ff(C.byref(d), _stream())), "rc_ff_kernel<320>" + ("+proj_out" if wpo is not None else ""),
              M, 320, 3 * int(inner), fl, True,
              alg_bytes=2.0 * ((4 if wpo is not None else 3) * M * 320 + 3 * 320 * int(inner) + (320 * 320 if wpo is not None else 0)))
    return out


def rc_front(x, coef, win, wqkv, rows_per_batch, ln_eps, y=None, qk=None, vt=None, dbg=0):
    """GroupNorm (coefficients) + proj_in + LayerNorm1 + q | k | v in one launch; see tg_rc_front.  Returns (y, qk, vt, ldt)."""
    from ._lib import RcFrontDesc
    _need_cuda(x)
    M, Cc = x.shape
    assert Cc == 320 and x.stride(1) == 1 and coef.dtype == torch.float32
    B = M // rows_per_batch
    ldt = (rows_per_batch + 7) // 8 * 8
    if y is None:
        y = torch.empty(M, 320, dtype=x.dtype, device=x.device)
    if qk is None:
        qk = torch.empty(M, 640, dtype=x.dtype, device=x.device)
    if vt is None:
        vt = torch.empty(B, 320, ldt, dtype=x.dtype, device=x.device)
    d = RcFrontDesc()
    d.dtype = _dt(x)
    d.x, d.ldx, d.coef, d.win, d.wqkv = _ptr(x), int(x.stride(0)), _ptr(coef), _ptr(win), _ptr(wqkv)
    d.y, d.ldy, d.qk, d.ldqk, d.vt, d.ldt = _ptr(y), int(y.stride(0)), _ptr(qk), int(qk.stride(0)), _ptr(vt), int(vt.stride(1))
    d.M, d.rows_per_batch, d.ln_eps, d.dbg = int(M), int(rows_per_batch), float(ln_eps), int(dbg)
    fl = 2.0 * M * 320 * (320 + 960)
    _profiled(lambda: _lib.check(_lib.lib().tg_rc_front(C.byref(d), _stream())), "rc_front_kernel<320>", M, 960, 320, fl, False,
              alg_bytes=2.0 * (5 * M * 320 + 4 * 320 * 320))
    return y, qk, vt, int(vt.stride(1))


def skinny_gemm(x, wpk, N, *, ln=None, bias=None, act=ACT_NONE, res=None, out=None, segs=None, rows_per_batch=None):
    """A handful of rows against a fragment-packed weight (``weights_pack.skinny_pack``); see tg_skinny_gemm.  ``ln`` = (u fp32 [N], v fp32 [N], eps): x is the
    un-normalised stream.  ``segs``: up to three ``(tensor_or_pointer, ld, batch_stride, n_end, transposed)`` output segments (default: one plain [M, N] tensor,
    returned).  Output pointers in ``segs`` are raw addresses (``tensor.data_ptr() + byte offset``): the caller keeps the tensors alive."""
    from ._lib import SkinnyDesc
    _need_cuda(x)
    M, K = x.shape
    assert x.stride(1) == 1 and wpk.numel() == N * K, (x.shape, wpk.numel(), N)
    d = SkinnyDesc()
    d.dtype, d.x, d.ldx, d.wpk, d.M, d.N, d.K = _dt(x), _ptr(x), int(x.stride(0)), _ptr(wpk), int(M), int(N), int(K)
    if ln is not None:
        d.ln, d.ln_u, d.ln_v, d.ln_eps = 1, _ptr(ln[0]), _ptr(ln[1]), float(ln[2])
    d.bias, d.act = _ptr(bias), int(act)
    if res is not None:
        d.res, d.ldres = _ptr(res), int(res.stride(0))
    if segs is None:
        if out is None:
            out = torch.empty(M, N, dtype=x.dtype, device=x.device)
        segs = [(out.data_ptr(), int(out.stride(0)), 0, N, 0)]
        rows_per_batch = M
    d.nseg, d.rows_per_batch = len(segs), int(rows_per_batch)
    for i, (ptr, ld, bs, n_end, tr) in enumerate(segs):
        d.seg[i].ptr, d.seg[i].ld, d.seg[i].batch_stride, d.seg[i].n_end, d.seg[i].transposed = int(ptr), int(ld), int(bs), int(n_end), int(tr)
    _lib.check(_lib.lib().tg_skinny_gemm(C.byref(d), _stream()))
    return out


def conv3x3(x, w_packed, batch, in_h, in_w, cin, *, x1=None, c1=0, stride=1, upsample=False, bias=None, pad_mode=0, **kw):
    """3x3 pad-1 convolution as implicit GEMM over token-major x [batch*in_h*in_w, cin] (+ optional concat x1).
    w_packed: [cout, 9*(cin+c1)] tap-major.  ``pad_mode=1``: zero padding on the bottom / right edge only (the VAE
    encoder's ``Downsample2D(padding=0)``: ``F.pad(x, (0, 1, 0, 1))`` then a stride-2 conv)."""
    if upsample:
        oh, ow = 2 * in_h, 2 * in_w
    else:
        pad2 = 1 if pad_mode == 1 else 2
        oh, ow = (in_h + pad2 - 3) // stride + 1, (in_w + pad2 - 3) // stride + 1
    N = w_packed.shape[0]
    K = 9 * (cin + c1)
    return gemm(x, w_packed, batch * oh * ow, N, K, mode=1, a1=x1, c0=cin, c1=c1,
                conv=(batch, in_h, in_w, oh, ow, stride, 1 if upsample else 0), bias=bias, pad_mode=pad_mode, **kw)


def attention(q, q_ld, q_bs, k0, k0_ld, k0_bs, vt0, vt0_ld, vt0_bs, len0, batch, heads, head_dim, n_q, scale,
              out, out_ld, out_bs, k1=None, k1_ld=0, k1_bs=0, vt1=None, vt1_ld=0, vt1_bs=0, len1=0, w1=0.0, causal=False, w1_dev=None,
              mask=None):
    """``w1_dev``: fp32 device scalar read by the kernel instead of the launch constant ``w1`` (graph-replayable IP scale).
    ``mask``: additive score bias, fp32 [Bm, Hm, Qm, len0] with Bm in {1, batch}, Hm in {1, heads}, Qm in {1, n_q} (size-1 dims broadcast)."""
    _need_cuda(q)
    d = AttnDesc()
    d.dtype = _dt(q)
    d.batch, d.heads, d.head_dim, d.n_q = int(batch), int(heads), int(head_dim), int(n_q)
    d.q, d.q_ld, d.q_bs = _ptr(q), int(q_ld), int(q_bs)
    d.k0, d.k0_ld, d.k0_bs = _ptr(k0), int(k0_ld), int(k0_bs)
    d.vt0, d.vt0_ld, d.vt0_bs = _ptr(vt0), int(vt0_ld), int(vt0_bs)
    d.len0 = int(len0)
    d.k1, d.k1_ld, d.k1_bs = _ptr(k1), int(k1_ld), int(k1_bs)
    d.vt1, d.vt1_ld, d.vt1_bs = _ptr(vt1), int(vt1_ld), int(vt1_bs)
    d.len1 = int(len1)
    d.scale, d.w1 = float(scale), float(w1)
    d.out, d.out_ld, d.out_bs = _ptr(out), int(out_ld), int(out_bs)
    d.causal = 1 if causal else 0
    if w1_dev is not None:
        if w1_dev.dtype != torch.float32 or not w1_dev.is_cuda:
            raise RuntimeError("attention: w1_dev must be an fp32 device tensor")
        d.w1_dev = w1_dev.data_ptr()
    if mask is not None:
        if mask.dtype != torch.float32 or not mask.is_cuda or mask.ndim != 4 or not mask.is_contiguous() or mask.shape[3] != len0:
            raise RuntimeError("attention: mask must be a contiguous fp32 device tensor [Bm, Hm, Qm, len0]")
        bm, hm, qm, _ = mask.shape
        if bm not in (1, batch) or hm not in (1, heads) or qm not in (1, n_q):
            raise RuntimeError(f"attention: mask shape {tuple(mask.shape)} does not broadcast to [{batch}, {heads}, {n_q}, {len0}]")
        d.mask = mask.data_ptr()
        d.mask_bs = hm * qm * len0 if bm > 1 else 0
        d.mask_hs = qm * len0 if hm > 1 else 0
        d.mask_qs = len0 if qm > 1 else 0
    if _gemm_profile is None:
        _lib.check(_lib.lib().tg_attention(C.byref(d), _stream()))
        return out
    # profiling mode (bench.py roofline leg): HIP events around this launch; algorithmic flops = QK^T + PV of both segments, no padding
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(_lib.lib().tg_attention(C.byref(d), _stream()))
    e1.record()
    keys = int(len0) + int(len1)
    _gemm_profile.append(dict(kernel=f"attention_kernel<d{int(head_dim)},{'self' if int(len0) == int(n_q) else 'cross'}>", splits=1,
                              M=int(batch) * int(n_q), N=int(heads) * int(head_dim), K=keys,
                              flops=4.0 * batch * heads * n_q * keys * head_dim, events=(e0, e1), has_res=False,
                              n_out=int(heads) * int(head_dim), attention=True, batch=int(batch)))
    return out


def attn_probs(q, q_ld, q_bs, k, k_ld, k_bs, batch, b0, heads, head_dim, n_q, length, scale, tokens=None):
    """fp32 [batch-b0, heads, n_q, n_tokens] softmax probabilities (save_attn_to_dict side channel)."""
    nt = length if tokens is None else int(tokens.numel())
    out = torch.empty((batch - b0, heads, n_q, nt), dtype=torch.float32, device=q.device)
    _lib.check(_lib.lib().tg_attn_probs(_dt(q), batch, b0, heads, head_dim, n_q, _ptr(q), q_ld, q_bs, _ptr(k), k_ld, k_bs,
                                        length, float(scale), _ptr(tokens), nt, _ptr(out), _stream()))
    return out


def groupnorm(x0, batch, hw, groups, eps, gamma, beta, silu=False, x1=None, out=None):
    _need_cuda(x0)
    c0 = x0.shape[-1]
    c1 = x1.shape[-1] if x1 is not None else 0
    L = _lib.lib()
    if out is None:
        out = torch.empty((batch * hw, c0 + c1), dtype=x0.dtype, device=x0.device)
    scratch = workspace(L.tg_groupnorm_scratch_bytes(batch, hw, groups), x0.device)
    _lib.check(L.tg_groupnorm(_dt(x0), _ptr(x0), _ptr(x1), c0, c1, batch, hw, groups, float(eps), _ptr(gamma), _ptr(beta),
                              1 if silu else 0, _ptr(out), _ptr(scratch), _stream()))
    return out


def groupnorm_coef(x0, batch, hw, groups, eps, gamma, beta, x1=None):
    """GroupNorm statistics only -> fp32 [batch, 2, C]: a = rstd * gamma, d = beta - mean * a (``conv3x3(..., a_coef=)``)."""
    _need_cuda(x0)
    c0 = x0.shape[-1]
    c1 = x1.shape[-1] if x1 is not None else 0
    L = _lib.lib()
    coef = torch.empty((batch, 2, c0 + c1), dtype=torch.float32, device=x0.device)
    scratch = workspace(L.tg_groupnorm_scratch_bytes(batch, hw, groups), x0.device)
    _lib.check(L.tg_groupnorm_coef(_dt(x0), _ptr(x0), _ptr(x1), c0, c1, batch, hw, groups, float(eps), _ptr(gamma), _ptr(beta),
                                   _ptr(coef), _ptr(scratch), _stream()))
    return coef


def groupnorm_from_partials(gn, batch, hw, C_, eps, gamma, beta, x=None, silu=False, out=None):
    """GroupNorm whose partial sums a producer conv already wrote (``gemm(gn_out=...)``): ``x`` None -> the coefficients fp32 [batch, 2, C]
    (``groupnorm_coef``), else the normalised (+ SiLU) tensor (``groupnorm``); no statistics pass over the activation."""
    L = _lib.lib()
    part, nblk, groups = gn["partials"], int(gn["nblk"]), int(gn["groups"])
    if part.shape != (batch, nblk, groups, 2) or part.dtype != torch.float32 or not part.is_cuda:
        raise RuntimeError("groupnorm_from_partials: partials must be the fp32 [batch, nblk, groups, 2] tensor a producer conv wrote")
    dt = _dt(x) if x is not None else (0 if gamma.dtype == torch.bfloat16 else 1)
    if x is None:
        coef = torch.empty((batch, 2, C_), dtype=torch.float32, device=part.device)
        _lib.check(L.tg_groupnorm_from_partials(dt, None, int(C_), int(batch), int(hw), groups, float(eps), _ptr(gamma), _ptr(beta), 0, None,
                                                _ptr(coef), _ptr(part), nblk, _stream()))
        return coef
    _need_cuda(x)
    if out is None:
        out = torch.empty((batch * hw, C_), dtype=x.dtype, device=x.device)
    _lib.check(L.tg_groupnorm_from_partials(dt, _ptr(x), int(C_), int(batch), int(hw), groups, float(eps), _ptr(gamma), _ptr(beta), 1 if silu else 0,
                                            _ptr(out), None, _ptr(part), nblk, _stream()))
    return out


_slab_plans = {}


def conv3x3_takes_gn(dtype, batch, in_h, in_w, cin, c1, cout):
    """True when tg_gemm runs this stride-1 conv on the slab kernel (tg_gemm_plan kernel_kind 4), which can apply
    GroupNorm + SiLU to its input while staging it (``a_coef``)."""
    key = (dtype, batch, in_h, in_w, cin, c1, cout, os.environ.get("TG_GEMM_FLAGS"))
    hit = _slab_plans.get(key)
    if hit is None:
        L = _lib.lib()
        d = GemmDesc()
        d.dtype = 0 if dtype == torch.bfloat16 else 1
        d.mode = 1
        d.a0 = d.w = d.out = 16                      # plan only: pointers are not dereferenced, just non-NULL and aligned
        d.a1 = 16 if c1 else None
        d.c0, d.c1 = int(cin), int(c1)
        d.batch, d.in_h, d.in_w, d.out_h, d.out_w, d.stride, d.upsample = batch, in_h, in_w, in_h, in_w, 1, 0
        d.M, d.N, d.K = batch * in_h * in_w, int(cout), 9 * (cin + c1)
        d.ldc = int(cout)
        d.out_scale = 1.0
        kk = C.c_int32()
        hit = L.tg_gemm_plan(C.byref(d), None, None, None, C.byref(kk)) == 0 and kk.value == 4
        _slab_plans[key] = hit
    return hit


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    _need_cuda(x)
    rows, Cc = x.shape
    if out is None:
        out = torch.empty((rows, Cc), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().tg_layernorm(_dt(x), _ptr(x), rows, Cc, x.stride(0), float(eps), _ptr(gamma), _ptr(beta), _ptr(out),
                                       out.stride(0), _stream()))
    return out


def layernorm_stats(x, eps=1e-5):
    """fp32 [rows, 2] = (rstd, -rstd * mean) per row: the statistics half of ``layernorm`` (no normalised tensor), input of ``gemm(ln=...)``"""
    _need_cuda(x)
    rows, Cc = x.shape
    out = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().tg_layernorm_stats(_dt(x), _ptr(x), rows, Cc, x.stride(0), float(eps), _ptr(out), _stream()))
    return out


def geglu(x, out=None):
    rows, two_inner = x.shape
    inner = two_inner // 2
    if out is None:
        out = torch.empty((rows, inner), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().tg_geglu(_dt(x), _ptr(x), rows, inner, _ptr(out), _stream()))
    return out


def act(x, kind, out=None):
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_lib.lib().tg_act(_dt(x), _ptr(x), x.numel(), kind, _ptr(out), _stream()))
    return out


def add(a, b, out=None):
    if out is None:
        out = torch.empty_like(a)
    _lib.check(_lib.lib().tg_add(_dt(a), _ptr(a), _ptr(b), a.numel(), _ptr(out), _stream()))
    return out


def attention_bwd_supported(head_dim, n):
    """True when tg_attention_bwd (recompute-based self-attention reverse pass) takes the problem."""
    return head_dim % 8 == 0 and head_dim <= 64 and n % 8 == 0


def attention_bwd(q, k, v, dout, batch, n, heads, head_dim, scale):
    """Self-attention reverse pass without materialised probabilities: q, k, v, dout [batch * n, heads * head_dim] (contiguous rows) ->
    (dq, dk, dv) in the same layout.  Three launches of one kernel (+ three tg_transpose launches for the streamed K^T / Q^T / dO^T tiles)."""
    _need_cuda(q)
    inner = heads * head_dim
    for t in (q, k, v, dout):
        if t.shape != (batch * n, inner) or not t.is_contiguous() or t.dtype != q.dtype:
            raise RuntimeError("attention_bwd: q, k, v, dout must be contiguous [batch * n, heads * head_dim] tensors of one dtype")
    L = _lib.lib()
    qt, kt, dot = transpose(q, batch, n, inner), transpose(k, batch, n, inner), transpose(dout, batch, n, inner)
    stats = torch.empty((batch, heads, n, 2), dtype=torch.float32, device=q.device)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    d = _lib.AttnBwdDesc()
    d.dtype, d.batch, d.heads, d.head_dim, d.n = _dt(q), batch, heads, head_dim, n
    d.q, d.k, d.v, d.dout = _ptr(q), _ptr(k), _ptr(v), _ptr(dout)
    d.ld, d.bs = inner, n * inner
    d.qt, d.kt, d.doutt = _ptr(qt), _ptr(kt), _ptr(dot)
    d.t_ld, d.t_bs = n, inner * n
    d.stats = _ptr(stats)
    d.dq, d.dk, d.dv = _ptr(dq), _ptr(dk), _ptr(dv)
    d.scale = float(scale)
    _lib.check(L.tg_attention_bwd(C.byref(d), _stream()))
    return dq, dk, dv


def attention_bwd_cross(q, dout, k, v, batch, n_q, n_k, heads, head_dim, scale, ds_scale, extra=None):
    """dQ of one softmax segment of cross-attention (constant K / V): q, dout [batch * n_q, inner], k, v [batch * n_k, inner];
    ``extra`` fp32 [batch, heads, n_q, n_k] = d loss / d P or None.  Two launches (row statistics, dQ) for all (item, head) pairs."""
    _need_cuda(q)
    inner = heads * head_dim
    lp = (n_k + 7) // 8 * 8
    kt = torch.zeros((batch, inner, lp), dtype=q.dtype, device=q.device)
    kt[:, :, :n_k] = k.reshape(batch, n_k, inner).transpose(1, 2)
    stats = torch.empty((batch, heads, n_q, 2), dtype=torch.float32, device=q.device)
    dq = torch.empty_like(q)
    d = _lib.AttnBwdCrossDesc()
    d.dtype, d.batch, d.heads, d.head_dim, d.n_q, d.n_k = _dt(q), batch, heads, head_dim, n_q, n_k
    d.q, d.dout, d.q_ld, d.q_bs = _ptr(q), _ptr(dout), inner, n_q * inner
    d.k, d.v, d.k_ld, d.k_bs = _ptr(k), _ptr(v), inner, n_k * inner
    d.kt, d.t_ld, d.t_bs = _ptr(kt), lp, inner * lp
    if extra is not None:
        if extra.dtype != torch.float32 or extra.shape != (batch, heads, n_q, n_k) or not extra.is_contiguous():
            raise RuntimeError("attention_bwd_cross: extra must be a contiguous fp32 [batch, heads, n_q, n_k] tensor")
        d.extra, d.extra_ld = _ptr(extra), n_k
    d.stats, d.dq = _ptr(stats), _ptr(dq)
    d.scale, d.ds_scale = float(scale), float(ds_scale)
    _lib.check(_lib.lib().tg_attention_bwd_cross(C.byref(d), _stream()))
    return dq


def transpose(src, batch, rows, cols, out=None):
    """out[b, c, r] = src[b, r, c]"""
    if out is None:
        out = torch.empty((batch, cols, rows), dtype=src.dtype, device=src.device)
    _lib.check(_lib.lib().tg_transpose(_dt(src), _ptr(src), batch, rows, cols, _ptr(out), _stream()))
    return out


def conv1x1_nchw(x, weight, bias, in_scale=1.0):
    """fp32 NCHW thin 1x1 conv (VAE post_quant_conv): x [B, cin, h, w], weight [cout, cin] fp32"""
    _need_cuda(x)
    B, cin, h, w = x.shape
    cout = weight.shape[0]
    out = torch.empty((B, cout, h, w), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().tg_conv1x1_nchw(_ptr(x), B, cin, cout, h * w, _ptr(weight), _ptr(bias), float(in_scale), _ptr(out), _stream()))
    return out


def softmax_rows(x, scale=1.0, out=None):
    """row softmax of a 2-D [rows, cols] tensor (cols % 8 == 0); ``out`` may alias ``x``"""
    _need_cuda(x)
    rows, cols = x.shape
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_lib.lib().tg_softmax_rows(_dt(x), _ptr(x), rows, cols, x.stride(0), float(scale), _ptr(out), out.stride(0), _stream()))
    return out


_SRC = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}


def conv_in(sample, weight_packed, bias, cout, dtype):
    """sample NCHW (bf16/f16/f32) -> token-major [B*h*w, cout] in ``dtype``"""
    _need_cuda(sample)
    B, cin, h, w = sample.shape
    out = torch.empty((B * h * w, cout), dtype=dtype, device=sample.device)
    _lib.check(_lib.lib().tg_conv_in(_dt(out), _ptr(sample), _SRC[sample.dtype], B, cin, h, w, _ptr(weight_packed), _ptr(bias),
                                     cout, _ptr(out), _stream()))
    return out


def conv_out_takes_gn(cin, h, w, cout):
    """True when conv_out runs on the matrix-core kernel, which can apply conv_norm_out + SiLU while it stages its window (``conv_out(..., coef=)``)."""
    return bool(_lib.lib().tg_conv_out_takes_coef(int(cin), int(h), int(w), int(cout)))


def conv_out(x, weight_packed, bias, batch, h, w, cout, out_dtype, coef=None, silu=True):
    """token-major [B*h*w, cin] -> NCHW [B, cout, h, w]; ``coef`` (groupnorm_coef of conv_norm_out): x is the raw block output, GroupNorm (+SiLU) applied
    inside the launch (only where conv_out_takes_gn)."""
    cin = x.shape[-1]
    out = torch.empty((batch, cout, h, w), dtype=out_dtype, device=x.device)
    if coef is not None:
        _lib.check(_lib.lib().tg_conv_out_gn(_dt(x), _ptr(x), _ptr(coef), 1 if silu else 0, batch, cin, h, w, _ptr(weight_packed), _ptr(bias), cout,
                                             _ptr(out), 1 if out_dtype == torch.float32 else 0, _stream()))
        return out
    _lib.check(_lib.lib().tg_conv_out(_dt(x), _ptr(x), batch, cin, h, w, _ptr(weight_packed), _ptr(bias), cout, _ptr(out),
                                      1 if out_dtype == torch.float32 else 0, _stream()))
    return out


def timestep_embedding(t_dev, rows, dim, flip_sin_to_cos, freq_shift, dtype, t_stride=0, index=None):
    out = torch.empty((rows, dim), dtype=dtype, device=t_dev.device)
    _lib.check(_lib.lib().tg_timestep_embedding(_dt(out), _ptr(t_dev), _ptr(index), t_stride, rows, dim, 1 if flip_sin_to_cos else 0,
                                                float(freq_shift), _ptr(out), dim, _stream()))
    return out


def step_epilogue(noise_pred, latents, guidance_scale, coef, step_idx, *, has_cfg=True, advance=True, prediction_type=0, frozen=None,
                  frozen_mask=None, frozen_steps=0, history=None, model_in=None):
    n_img = latents.shape[0]
    chw = latents[0].numel()
    hw = latents.shape[-1] * latents.shape[-2]
    mask_per_img = 1 if (frozen_mask is not None and frozen_mask.numel() == n_img * hw and n_img > 1) else 0
    mi_dt = -1 if model_in is None else _SRC[model_in.dtype]
    _lib.check(_lib.lib().tg_step_epilogue(_ptr(noise_pred), _ptr(latents), n_img, chw, hw, 1 if has_cfg else 0, float(guidance_scale), _ptr(coef),
                                           _ptr(step_idx), 1 if advance else 0, prediction_type, _ptr(frozen), _ptr(frozen_mask),
                                           mask_per_img, int(frozen_steps), _ptr(history), _ptr(model_in), mi_dt, _stream()))


def blend_latents(bg, fg, mask, ratio, sigma=1.0, storage_dtype=None):
    """fp32 in / out; ``storage_dtype`` torch.float16 / torch.bfloat16: reproduce the half-precision roundings of the
    reference expression (its latents are ``unet.dtype`` tensors)"""
    out = torch.empty_like(bg)
    hw = bg.shape[-1] * bg.shape[-2]
    sd = {None: -1, torch.float32: -1, torch.bfloat16: _lib.TG_BF16, torch.float16: _lib.TG_F16}[storage_dtype]
    _lib.check(_lib.lib().tg_blend_latents(_ptr(bg), _ptr(fg), _ptr(mask), bg.numel() // hw, hw, float(ratio), float(sigma),
                                           sd, _ptr(out), _stream()))
    return out


def gaussian_sample(moments, noise, scale=1.0):
    """moments fp32 [B, 2C, h, w] -> scale * (mean + std * noise) fp32 [B, C, h, w]; ``noise`` None = the mode"""
    B, C2, h, w = moments.shape
    out = torch.empty((B, C2 // 2, h, w), dtype=torch.float32, device=moments.device)
    _lib.check(_lib.lib().tg_gaussian_sample(_ptr(moments), _ptr(noise), B, C2 // 2, h * w, float(scale), _ptr(out), _stream()))
    return out


def add_noise(x0, noise, ca, cb):
    """out[s] = ca[s] * x0 + cb[s] * noise for a table of `steps` coefficients (fp32)"""
    steps = ca.numel()
    out = torch.empty((steps, *x0.shape), dtype=torch.float32, device=x0.device)
    _lib.check(_lib.lib().tg_add_noise(_ptr(x0), _ptr(noise), _ptr(ca), _ptr(cb), steps, x0.numel(), _ptr(out), _stream()))
    return out


def shift(src, dx, dy):
    h, w = src.shape[-2:]
    out = torch.empty_like(src)
    _lib.check(_lib.lib().tg_shift(_ptr(src), src.numel() // (h * w), h, w, int(dx), int(dy), _ptr(out), _stream()))
    return out


def masked_compose_(dst, src, mask):
    hw = dst.shape[-1] * dst.shape[-2]
    _lib.check(_lib.lib().tg_masked_compose(_ptr(dst), _ptr(src), _ptr(mask), dst.numel() // hw, hw, _stream()))
    return dst


def guidance_topk(attn, token, mask, k_fg, k_bg, fg_w, bg_w, scale, out, grad=None):
    for t in (attn, mask, out, grad):
        if t is not None:
            _need_cuda(t)
    heads, hw, n_tok = attn.shape
    _lib.check(_lib.lib().tg_guidance_topk(_ptr(attn), heads, hw, n_tok, int(token), _ptr(mask), int(k_fg), int(k_bg),
                                           float(fg_w), float(bg_w), float(scale), _ptr(out), _ptr(grad), _stream()))


def guidance_ref(attn, token, ref, mask, eps, scale, out, grad=None):
    """attn fp32 [heads, hw, tokens]; ref fp32 [heads, hw] contiguous"""
    for t in (attn, ref, mask, out, grad):
        if t is not None:
            _need_cuda(t)
    heads, hw, n_tok = attn.shape
    if ref.dtype != torch.float32 or ref.numel() != heads * hw or not ref.is_contiguous():
        raise RuntimeError("guidance_ref: ref must be a contiguous fp32 [heads, hw] column")
    _lib.check(_lib.lib().tg_guidance_ref(_ptr(attn), heads, hw, n_tok, int(token), _ptr(ref), _ptr(mask), float(eps),
                                          float(scale), _ptr(out), _ptr(grad), _stream()))


def guidance_ratio(attn, token, mask, scale, out, grad=None):
    for t in (attn, mask, out, grad):
        if t is not None:
            _need_cuda(t)
    heads, hw, n_tok = attn.shape
    _lib.check(_lib.lib().tg_guidance_ratio(_ptr(attn), heads, hw, n_tok, int(token), _ptr(mask), float(scale), _ptr(out),
                                            _ptr(grad), _stream()))


# ---- input-gradient kernels (latent_backward_guidance) ------------------------------------------------------------------
def groupnorm_bwd(x, dy, batch, hw, groups, eps, gamma, beta, silu=False):
    _need_cuda(x)
    dx = torch.empty_like(x)
    L = _lib.lib()
    scratch = workspace(L.tg_groupnorm_bwd_scratch_bytes(int(batch), int(hw), int(groups)), x.device)
    _lib.check(L.tg_groupnorm_bwd(_dt(x), _ptr(x), _ptr(dy), int(batch), int(hw), x.shape[-1], int(groups), float(eps), _ptr(gamma),
                                  _ptr(beta), 1 if silu else 0, _ptr(dx), _ptr(scratch), _stream()))
    return dx


def layernorm_bwd(x, dy, gamma, eps=1e-5):
    _need_cuda(x)
    rows, Cc = x.shape
    dx = torch.empty_like(x)
    _lib.check(_lib.lib().tg_layernorm_bwd(_dt(x), _ptr(x), _ptr(dy), rows, Cc, float(eps), _ptr(gamma), _ptr(dx), _stream()))
    return dx


def geglu_bwd(h, dg):
    rows, two_inner = h.shape
    dh = torch.empty_like(h)
    _lib.check(_lib.lib().tg_geglu_bwd(_dt(h), _ptr(h), _ptr(dg), rows, two_inner // 2, _ptr(dh), _stream()))
    return dh


def softmax_bwd_rows(probs, dprobs, length, scale, ld_out, extra=None, want_probs=False):
    """probs fp32 or storage dtype [rows, >= length] (row pitch probs.stride(0)), dprobs storage dtype [rows, >= length];
    -> dS [rows, ld_out] (and P in the storage dtype, same shape, when ``want_probs``)"""
    rows = probs.shape[0]
    if probs.dtype != torch.float32 and probs.dtype != dprobs.dtype:
        raise RuntimeError("softmax_bwd_rows: probs must be fp32 or the storage dtype")
    ds = torch.empty((rows, ld_out), dtype=dprobs.dtype, device=dprobs.device)
    pout = torch.empty_like(ds) if want_probs else None
    _lib.check(_lib.lib().tg_softmax_bwd_rows(_dt(dprobs), _ptr(probs), 1 if probs.dtype == torch.float32 else 0, probs.stride(0),
                                              _ptr(dprobs), dprobs.stride(0), _ptr(extra),
                                              extra.stride(0) if extra is not None else 0, rows, int(length), float(scale), _ptr(ds),
                                              _ptr(pout), int(ld_out), _stream()))
    return (ds, pout) if want_probs else ds


def sumpool2x2(du, batch, h, w):
    """du token-major [batch * 2h * 2w, C] -> [batch * h * w, C]"""
    Cc = du.shape[-1]
    out = torch.empty((batch * h * w, Cc), dtype=du.dtype, device=du.device)
    _lib.check(_lib.lib().tg_sumpool2x2(_dt(du), _ptr(du), int(batch), int(h), int(w), Cc, _ptr(out), _stream()))
    return out


def capturing_now():
    """True while the current stream is being captured into a hipGraph"""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


class GuidanceBatch:
    """Collects the terms of one ``compute_ca_lossv3`` call (one per attention map x object x token position) and evaluates
    them with ONE ``tg_guidance_plan_run`` launch pair (+ an in-order fold) instead of one dependent launch each.

    Round 3: the device-side item table holds SLOT indices, not pointers; the call's tensors (maps, gradients, masks, reference
    columns) are passed as kernel arguments.  The table therefore only depends on (kinds, shapes, token positions, top-k sizes,
    weights, which tensor goes with which) and is cached on the device, keyed by exactly that — a steady-state call makes NO host ->
    device copy (the round-2 path built and uploaded a pointer table per call: most of its 1.3 ms at BASELINE configs[3]) and can
    be captured in a hipGraph.

    Terms whose gradient columns would collide (same grad tensor and token in one call: two objects sharing a token position, or a
    reference-attention term next to the box term of the same key and token) are deferred to a following launch, so the
    order of additions into the loss is LAUNCH-MAJOR, item order within each launch — fixed and deterministic for a given
    call, identical to the per-term launches whenever no column collides (the shipped flow), one fp32 re-association otherwise."""
    KIND_TOPK, KIND_RATIO, KIND_REF = 0, 1, 2
    _tables = {}            # (device index, item signature) -> (device table tensor, n_items, max_hw_topk, max_heads)
    _TABLES_MAX = 128
    _pinned = {}            # entries a stream CAPTURE has read: a captured hipGraph bakes the table's device address, so they are never
                            # evicted (a 64-byte row per term; ADVICE r3: the FIFO above recycled tables under still-cached graphs)

    def __init__(self, device):
        self.device = device
        self.items = []

    def add(self, kind, attn, token, mask, scale, grad=None, ref=None, k_fg=0, k_bg=0, fg_w=0.0, bg_w=0.0, eps=0.0):
        heads, hw, n_tok = attn.shape
        for t in (attn, grad, mask, ref):             # raw device pointers go to the kernel: a host tensor would fault the GPU
            if t is not None:
                _need_cuda(t)
                if t.dtype != torch.float32 or not t.is_contiguous():
                    raise RuntimeError("guidance: attention maps, gradients, masks and reference columns must be contiguous fp32")
        if not (0 <= int(token) < n_tok):
            raise RuntimeError(f"guidance: token position {token} outside the {n_tok} text tokens")
        if kind == self.KIND_TOPK and not (1 <= k_fg <= hw and 1 <= k_bg <= hw):
            raise RuntimeError("guidance: top-k size out of range")
        if mask.numel() != hw or (ref is not None and ref.numel() != heads * hw) or (grad is not None and grad.numel() != attn.numel()):
            raise RuntimeError("guidance: mask / reference / gradient shape does not match the attention map")
        self.items.append(dict(kind=int(kind), attn=attn, grad=grad, mask=mask, ref=ref, heads=int(heads), hw=int(hw), n_tok=int(n_tok),
                               token=int(token), k_fg=int(k_fg), k_bg=int(k_bg), fg_w=float(fg_w), bg_w=float(bg_w), scale=float(scale),
                               eps=float(eps)))

    def _launch(self, now, loss):
        slots, index = [], {}

        def slot(t):
            if t is None:
                return -1
            key = t.data_ptr()
            if key not in index:
                index[key] = len(slots)
                slots.append(t)
            return index[key]
        rows = []
        for it in now:
            rows.append((slot(it["attn"]), slot(it["grad"]), slot(it["mask"]), slot(it["ref"]), it["heads"], it["hw"], it["n_tok"], it["token"],
                         it["kind"], it["k_fg"], it["k_bg"], it["fg_w"], it["bg_w"], it["scale"], it["eps"]))
        if len(slots) > _lib.GUIDANCE_MAX_SLOTS:
            half = len(now) // 2                      # more distinct tensors than argument slots: two launches (item order is kept)
            self._launch(now[:half], loss)
            self._launch(now[half:], loss)
            return
        key = (self.device.index, tuple(rows))
        capturing = capturing_now()
        hit = GuidanceBatch._pinned.get(key) or GuidanceBatch._tables.get(key)
        if hit is None:
            if capturing:
                raise RuntimeError("guidance: this term layout has no device table yet and a stream capture is in progress (the upload "
                                   "cannot be captured): run the same call once eagerly before capturing it")
            arr = (_lib.GuidancePItem * len(rows))()
            for a, r in zip(arr, rows):
                (a.attn_slot, a.grad_slot, a.mask_slot, a.ref_slot, a.heads, a.hw, a.n_tok, a.token, a.kind, a.k_fg, a.k_bg,
                 a.fg_w, a.bg_w, a.scale, a.eps) = r
            table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)     # once per signature
            hit = (table, len(rows), max([r[5] for r in rows if r[8] == self.KIND_TOPK] + [0]), max(r[4] for r in rows))
            if len(GuidanceBatch._tables) >= GuidanceBatch._TABLES_MAX:
                GuidanceBatch._tables.pop(next(iter(GuidanceBatch._tables)))
            GuidanceBatch._tables[key] = hit
        if capturing:
            GuidanceBatch._pinned[key] = hit
        table, n, max_hw, max_heads = hit
        ptrs = (C.c_void_p * len(slots))(*[t.data_ptr() for t in slots])
        head_terms = torch.empty(n * max_heads, dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().tg_guidance_plan_run(table.data_ptr(), n, max_hw, max_heads, ptrs, len(slots), head_terms.data_ptr(), _ptr(loss),
                                                   _stream()))

    def flush(self, loss):
        """loss: fp32 device tensor [1], accumulated in place (launch-major, item order within a launch)"""
        items, self.items = self.items, []
        while items:
            seen, now, later = set(), [], []
            for it in items:
                col = (it["grad"].data_ptr(), it["token"]) if it["grad"] is not None else None
                if col is not None and col in seen:
                    later.append(it)              # keeps item order within each launch; collisions go to the next one
                else:
                    if col is not None:
                        seen.add(col)
                    now.append(it)
            self._launch(now, loss)
            items = later
        # eager launches pin nothing beyond the last launch: every tensor is used on the current stream only, and the caching allocator
        # hands a freed block back to that same stream (stream-ordered reuse).  Under a stream capture the item table is pinned above,
        # the box masks by guidance._box_mask, and the per-call tensors (maps, gradients, head terms) live in the graph's private pool.
        return loss

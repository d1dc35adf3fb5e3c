"""Host side of the row-chain kernels (csrc/tg_rowchain.hip): weight preprocessing for ``tg_rc_xattn`` — norm2 + attn2 + residual of a
first-level ``BasicTransformerBlock`` (models/attention.py:206-224; ``IPAttnProcessor`` ip_adapter/attention_processor.py:445-529) in one
launch for SD-1.5's geometry (320 channels = 8 heads x 40).  Load-time plumbing only: permutations of ``to_q`` rows / ``to_out`` columns
so that the kernel's register-resident q / O pieces line up with its MFMA operands, the LayerNorm fold of ``pack_ln_linear``, and the
softmax scale (x log2 e: the kernel uses exp2) folded into ``to_q``."""
import math
import os

import torch

from . import ops
from .weights_pack import rc_pack, rc_pack_tiles

# TG_RC: 0 = the row-chain kernels are never selected (old-path A/B), 1 = default.  TG_RC_MIN_ROWS: below this many token rows the
# LDS-tiled GEMMs / the three-launch cross-attention stay (a row-chain workgroup is a long serial chain: it needs a full chip of them)
ENABLED = os.environ.get("TG_RC", "1") != "0"
# dev A/B bits: 0 tg_rc_linear swaps (K = 320), 1 tg_rc_xattn, 2 tg_rc_ff, 3 tg_rc_front
MODE = int(os.environ.get("TG_RC_MODE", "15"))
MIN_ROWS = int(os.environ.get("TG_RC_MIN_ROWS", "8192"))
# rc_linear / rc_front / rc_ff are one long serial chain per workgroup (a lone rc_ff workgroup needs ~95 us): below one full round of 128-token
# workgroups (256 CUs x 128 rows) the LDS-tiled launches are as fast or faster (8192 rows: rc_ff 95 us vs 64 us, rc_front 51 + 16 vs 59 us)
MIN_ROWS_CHAIN = int(os.environ.get("TG_RC_MIN_ROWS_CHAIN", "32768"))
TRACE = os.environ.get("TG_RC_TRACE") == "1"


def trace(*a):
    if TRACE:
        print("[tg_rc]", *a, flush=True)

HEADS, DH, C = 8, 40, 320


def _q_slot_channels():
    """q channel (head * 40 + d) held by slot n' = 64 c + 32 hi + 8 i + j of the kernel's 20 q k-steps (k-step 4 c + i, lane half hi, element j)"""
    out = torch.empty(C, dtype=torch.long)
    for n in range(C):
        c, hi, i, j = n // 64, (n // 32) & 1, (n // 8) & 3, n & 7
        s = 4 * c + i
        m, w5 = s // 5, s % 5
        if w5 == 0:
            head, d = 2 * m, 8 * hi + j
        elif w5 == 1:
            head, d = 2 * m, 16 + 8 * hi + j
        elif w5 == 2:
            head, d = (2 * m, 32 + j) if hi == 0 else (2 * m + 1, j)
        elif w5 == 3:
            head, d = 2 * m + 1, 8 + 8 * hi + j
        else:
            head, d = 2 * m + 1, 24 + 8 * hi + j
        out[n] = head * DH + d
    assert sorted(out.tolist()) == list(range(C))
    return out


def _o_slot_channels():
    """attention-output channel held by slot n' of the kernel's 20 O k-steps (what ``to_out`` contracts over)"""
    out = torch.empty(C, dtype=torch.long)
    for n in range(C):
        c, hi, i, j = n // 64, (n // 32) & 1, (n // 8) & 3, n & 7
        t = 4 * c + i
        m, t5 = t // 5, t % 5
        if t5 in (0, 1):
            head, d = 2 * m, 16 * t5 + 8 * (j >> 2) + 4 * hi + (j & 3)
        elif t5 == 2:
            head, d = (2 * m, 32 + 4 * hi + j) if j < 4 else (2 * m + 1, 32 + 4 * hi + j - 4)
        else:
            head, d = 2 * m + 1, 16 * (t5 - 3) + 8 * (j >> 2) + 4 * hi + (j & 3)
        out[n] = head * DH + d
    assert sorted(out.tolist()) == list(range(C))
    return out


_QS, _OS = None, None



def pack_xattn_q(wq, bq, gamma, beta, scale):
    """``to_q`` [320, 320] (+ bias) behind LayerNorm(gamma, beta), times softmax scale * log2(e) -> rc chunk stream (uint8).
    q' = rstd * (x W'^T - mean u) + v with W' = s W gamma rounded to the storage dtype, u = row sums of the ROUNDED W', v = s (W beta + b)."""
    global _QS
    if _QS is None:
        _QS = _q_slot_channels()
    s = float(scale) * math.log2(math.e)
    w32 = wq.detach().float()
    wp = (w32 * gamma.detach().float()[None, :] * s).to(wq.dtype)
    u = wp.float().sum(dim=1)
    v = w32 @ beta.detach().float()
    if bq is not None:
        v = v + bq.detach().float()
    v = v * s
    perm = _QS.to(wq.device)
    return rc_pack_tiles(wp[perm].contiguous(), v[perm], u[perm])


def pack_xattn_out(wo, bo):
    """``to_out[0]`` [320, 320] + bias -> rc chunk stream with the input columns in the kernel's O-piece order"""
    global _OS
    if _OS is None:
        _OS = _o_slot_channels()
    perm = _OS.to(wo.device)
    w2 = wo.detach()[:, perm].contiguous()
    return rc_pack_tiles(w2, bo.detach().float() if bo is not None else None)


def linear320(x2d, lin_weight, bias, res, owner, name, cached):
    """``x @ W^T + b (+ res)`` through ``tg_rc_linear`` when the shape pays — K = 320, rows >= MIN_ROWS_CHAIN, N % 64 == 0 —, else None.
    ``cached(owner, name, tensors, build)`` is the caller's packed-weight cache."""
    M, K = x2d.shape
    N = lin_weight.shape[0]
    if not ENABLED or x2d.stride(1) != 1 or lin_weight.shape[1] != K:
        return None
    ts = [lin_weight] + ([bias] if bias is not None else [])
    if K == 320 and (MODE & 1) and N % 64 == 0 and M >= MIN_ROWS_CHAIN:
        trace("linear320 yes", name, M, N)
        wpk = cached(owner, "rc_" + name, ts, lambda: rc_pack(lin_weight.detach(), bias.detach().float() if bias is not None else None))
        return ops.rc_linear(x2d, wpk, N, res=res)
    trace("linear_rc no", name, M, K, N)
    return None


def xattn_eligible(attn, C, B, N, L, T, dtype):
    inner = attn.to_q.weight.shape[0]
    return (ENABLED and (MODE & 2) and C == 320 and inner == 320 and int(attn.heads) == 8 and L == 77 and T in (0, 4, 16) and N % 128 == 0
            and B * N >= MIN_ROWS and dtype in (torch.bfloat16, torch.float16) and attn.to_out[0].weight.shape[0] == 320)


# ---- tg_rc_ff: norm3 + FeedForward (GEGLU) + residual (+ proj_out + residual) of a first-level block in one launch --------------------
def _ff_row_of():
    """MFMA row r of an FF1 tile -> offset of its hidden channel inside the 32-channel slice: accumulator register rho of lane half hi
    (row = (rho & 3) + 8 (rho >> 2) + 4 hi) holds hidden channel 16 hi + rho"""
    r = torch.arange(32)
    rho = (r & 3) + 4 * (r >> 3)
    hi = (r >> 2) & 1
    return 16 * hi + rho


def pack_ff(w1, b1, gamma, beta, w2, b2):
    """``GEGLU.proj`` [2 * inner, 320] (+ bias) behind LayerNorm(gamma, beta) and ``net.2`` [320, inner] (+ bias) -> the streams of
    ``tg_rc_ff`` (reference models/attention.py:226-236, 328-338; inner = 1280):
      * w1 stream: per 32-channel hidden slice j: 20 fragment blocks of the VALUE rows, 20 of the GATE rows (K order = the row layout of
        ``rc_pack``), then one 1-KiB page: fp32 bias_a[32], bias_g[32] in accumulator order, W * gamma folded, bias + W beta folded;
        the kernel normalises the rows itself ((x - mean) * rstd rounded to the storage dtype: what ``norm3`` hands to ``ff``), 3 KiB pad;
      * w2 stream: per slice j: 10 output tiles x 2 k-steps of ``net.2`` columns [32 j, 32 j + 32) (20 KiB);
      * b2: fp32 [320]."""
    from .weights_pack import _rc_maps
    inner = w2.shape[1]
    assert w1.shape == (2 * inner, C) and w2.shape == (C, inner) and inner % 32 == 0
    dev, dt = w1.device, w1.dtype
    g32, be32 = gamma.detach().float(), beta.detach().float()
    w1f = w1.detach().float()
    w1g = (w1f * g32[None, :]).to(dt)                                   # rounded once, like pack_ln_linear
    v1 = w1f @ be32 + (b1.detach().float() if b1 is not None else 0.0)   # [2 * inner]
    pi, kap = _rc_maps(C)
    kap = kap.to(dev)                                                   # [s, hi, j] -> input channel
    rowoff = _ff_row_of().to(dev)                                       # [r] -> hidden offset in the slice
    ns = inner // 32
    hid = 32 * torch.arange(ns, device=dev)[:, None] + rowoff[None, :]  # [slice, r]
    rows_a, rows_g = hid, inner + hid
    # frag[slice, which(a / g), s, hi, r, j] = w1g[row(slice, which, r), kap[s, hi, j]]
    rows = torch.stack([rows_a, rows_g], dim=1)                          # [slice, 2, r]
    frag = w1g[rows[:, :, None, None, :, None], kap[None, None, :, :, None, :]].contiguous()
    fb = frag.reshape(ns, -1).view(torch.uint8)                          # [slice, 40 KiB]
    page = torch.zeros(ns, 256, dtype=torch.float32, device=dev)
    acc_ch = (16 * torch.arange(2, device=dev)[:, None] + torch.arange(16, device=dev)[None, :]).reshape(-1)   # index 16 hi + rho -> hidden offset
    page[:, :32] = v1[(32 * torch.arange(ns, device=dev)[:, None] + acc_ch[None, :])]
    page[:, 32:64] = v1[inner + (32 * torch.arange(ns, device=dev)[:, None] + acc_ch[None, :])]
    s1 = torch.cat([fb, page.view(torch.uint8)], dim=1).reshape(-1)
    s1 = torch.cat([s1, torch.zeros(3 * 1024, dtype=torch.uint8, device=dev)])
    # w2: block (slice, t = 2 c + u, kk): lane (hi, r), element j = w2[64 c + pi(u, r), 32 slice + 16 hi + 8 kk + j]
    out_rows = (64 * torch.arange(C // 64)[:, None, None] + pi[None]).reshape(-1, 32).to(dev)      # [t, r]
    kcol = (32 * torch.arange(ns, device=dev)[:, None, None, None] + 16 * torch.arange(2, device=dev)[None, None, :, None]
            + 8 * torch.arange(2, device=dev)[None, :, None, None] + torch.arange(8, device=dev)[None, None, None, :])   # [slice, kk, hi, j]
    w2d = w2.detach()
    f2 = w2d[out_rows[None, :, None, None, :, None], kcol[:, None, :, :, None, :]].contiguous()    # [slice, t, kk, hi, r, j]
    s2 = f2.reshape(-1).view(torch.uint8)
    bb2 = (b2.detach().float() if b2 is not None else torch.zeros(C, device=dev)).contiguous()
    return s1.contiguous(), s2.contiguous(), bb2

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
MODE = int(os.environ.get("TG_RC_MODE", "3"))      # dev A/B: bit 0 = tg_rc_linear swaps, bit 1 = tg_rc_xattn
MIN_ROWS = int(os.environ.get("TG_RC_MIN_ROWS", "8192"))
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
    """``x @ W^T + b (+ res)`` for K = 320 through ``tg_rc_linear`` when the shape pays (rows >= MIN_ROWS, N % 64 == 0), else None.
    ``cached(owner, name, tensors, build)`` is the caller's packed-weight cache."""
    M, K = x2d.shape
    N = lin_weight.shape[0]
    if not ENABLED or not (MODE & 1) or K != 320 or N % 64 or M < MIN_ROWS or x2d.stride(1) != 1 or lin_weight.shape[1] != 320:
        trace("linear320 no", name, M, K, N)
        return None
    trace("linear320 yes", name, M, N)
    ts = [lin_weight] + ([bias] if bias is not None else [])
    wpk = cached(owner, "rc_" + name, ts, lambda: rc_pack(lin_weight.detach(), bias.detach().float() if bias is not None else None))
    return ops.rc_linear(x2d, wpk, N, res=res)


def xattn_eligible(attn, C, B, N, L, T, dtype):
    inner = attn.to_q.weight.shape[0]
    return (ENABLED and (MODE & 2) and C == 320 and inner == 320 and int(attn.heads) == 8 and L == 77 and T in (0, 4, 16) and N % 128 == 0
            and B * N >= MIN_ROWS and dtype in (torch.bfloat16, torch.float16) and attn.to_out[0].weight.shape[0] == 320)

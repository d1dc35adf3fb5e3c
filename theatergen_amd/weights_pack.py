"""Weight layout conversion (load-time plumbing, host side): diffusers state-dict tensors -> the layouts
the HIP kernels read.  Token-major activations want K-contiguous weights:
  Linear  [N, K]            -> unchanged
  Conv1x1 [N, C, 1, 1]      -> [N, C]
  Conv3x3 [N, C, 3, 3]      -> [N, 9*C] tap-major (ky, kx, c)  (implicit-GEMM K order of tg_gemm mode 1)
"""
import torch


def pack_conv3x3(w: torch.Tensor) -> torch.Tensor:
    n, c, kh, kw = w.shape
    assert kh == 3 and kw == 3
    return w.permute(0, 2, 3, 1).reshape(n, 9 * c).contiguous()


def pack_conv1x1(w: torch.Tensor) -> torch.Tensor:
    return w.reshape(w.shape[0], w.shape[1]).contiguous()


def pack_geglu(w: torch.Tensor, b: torch.Tensor = None):
    """GEGLU.proj weight [2*inner, C] = [a ; gate] (reference models/attention.py:328, 337) -> rows interleaved in groups
    of 32: [a(0:32) ; gate(0:32) ; a(32:64) ; gate(32:64) ; ...] so that one wave of tg_gemm holds a[c] and gate[c] of
    the same channel in the same lane (fused GEGLU epilogue).  ``inner`` must be a multiple of 32."""
    two_inner = w.shape[0]
    inner = two_inner // 2
    assert inner % 32 == 0
    a, g = w[:inner], w[inner:]
    wp = torch.stack([a.reshape(inner // 32, 32, -1), g.reshape(inner // 32, 32, -1)], dim=1).reshape(two_inner, -1).contiguous()
    bp = None
    if b is not None:
        bp = torch.stack([b[:inner].reshape(inner // 32, 32), b[inner:].reshape(inner // 32, 32)], dim=1).reshape(two_inner).contiguous()
    return wp, bp


def pack_ln_linear(w: torch.Tensor, bias, gamma: torch.Tensor, beta: torch.Tensor):
    """LayerNorm folded into the Linear that consumes it (tg_gemm ``ln_u`` / ``ln_v``):
        Linear(LayerNorm(x)) = rstd * (x W'^T - mean * u) + v,   W' = W * gamma  (storage dtype),
        u[n] = sum_k W'[n, k]  summed from the ROUNDED W' (so that x W'^T - mean * u = (x - mean) W'^T exactly),
        v[n] = sum_k beta[k] W[n, k] + bias[n]                     (fp32).
    One-off weight preprocessing at load / first use, like the conv repacks above; returns (W', u fp32, v fp32)."""
    w32 = w.detach().float()
    wp = (w32 * gamma.detach().float()[None, :]).to(w.dtype).contiguous()
    u = wp.float().sum(dim=1).contiguous()
    v = w32 @ beta.detach().float()
    if bias is not None:
        v = v + bias.detach().float()
    return wp, u, v.contiguous()

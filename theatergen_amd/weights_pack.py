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

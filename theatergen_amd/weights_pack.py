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

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


def pack_ln_linear(w: torch.Tensor, bias, gamma: torch.Tensor, beta: torch.Tensor, scale: float = 1.0):
    """LayerNorm folded into the Linear that consumes it (tg_gemm ``ln_u`` / ``ln_v``):
        Linear(LayerNorm(x)) = rstd * (x W'^T - mean * u) + v,   W' = W * gamma  (storage dtype),
        u[n] = sum_k W'[n, k]  summed from the ROUNDED W' (so that x W'^T - mean * u = (x - mean) W'^T exactly),
        v[n] = sum_k beta[k] W[n, k] + bias[n]                     (fp32).
    One-off weight preprocessing at load / first use, like the conv repacks above; returns (W', u fp32, v fp32).
    ``scale`` multiplies W and the bias in fp32 BEFORE the rounding (tg_xq_attn: softmax scale * log2(e) folded into to_q)."""
    w32 = w.detach().float() * float(scale)
    wp = (w32 * gamma.detach().float()[None, :]).to(w.dtype).contiguous()
    u = wp.float().sum(dim=1).contiguous()
    v = w32 @ beta.detach().float()
    if bias is not None:
        v = v + bias.detach().float() * float(scale)
    return wp, u, v.contiguous()


def _rc_maps(K: int):
    """index maps of the row-chain kernels (csrc/tg_rowchain.hip): output-row permutation of a 64-row chunk and the k order"""
    r = torch.arange(32)
    pi = torch.stack([32 * ((r >> 2) & 1) + 16 * u + 4 * (r >> 3) + (r & 3) for u in range(2)])      # [u, r] -> row within the chunk
    s = torch.arange(K // 16)[:, None, None]
    hi = torch.arange(2)[None, :, None]
    j = torch.arange(8)[None, None, :]
    kap = 64 * (s >> 2) + 32 * hi + 8 * (s & 3) + j                                                   # [s, hi, j] -> input channel
    return pi, kap


def rc_pack(w: torch.Tensor, v: torch.Tensor = None, u: torch.Tensor = None) -> torch.Tensor:
    """Linear weight [N, K] (N % 64 == 0, K % 64 == 0) -> the row-chain kernels' chunk stream (uint8): per 64-row chunk c
      * 2 MFMA tiles x K / 16 k-steps x (64 lanes x 8 elements) of weight fragments: block (u, s), lane (hi, r), element j =
        W[64 c + pi(u, r), kappa(s, hi, j)]  (straight LDS-DMA, linear conflict-free fragment reads), then
      * one 1-KiB vector page: fp32 v[64 c : 64 c + 64] (bias, or W beta + bias under the LayerNorm fold), fp32 u[64 c : 64 c + 64]
        (fold: row sums of the ROUNDED W gamma), zero padding."""
    n, k = w.shape
    assert n % 64 == 0 and k % 64 == 0, (n, k)
    pi, kap = _rc_maps(k)
    rows = (64 * torch.arange(n // 64)[:, None, None] + pi[None]).to(w.device)          # [c, u, r]
    kap = kap.to(w.device)
    # frag[c, u, s, hi, r, j]
    frag = w.detach()[rows[:, :, None, None, :, None], kap[None, None, :, :, None, :]].contiguous()
    fb = frag.reshape(n // 64, -1).view(torch.uint8)                                      # [c, 128 K bytes]
    page = torch.zeros(n // 64, 256, dtype=torch.float32, device=w.device)
    if v is not None:
        page[:, :64] = v.detach().float().reshape(n // 64, 64)
    if u is not None:
        page[:, 64:128] = u.detach().float().reshape(n // 64, 64)
    return torch.cat([fb, page.view(torch.uint8)], dim=1).contiguous().reshape(-1)


def rc_pack_tiles(w: torch.Tensor, v: torch.Tensor = None, u: torch.Tensor = None, page: bool = True) -> torch.Tensor:
    """As ``rc_pack`` but one stream element per 32-row MFMA TILE (tile t = 2 c + u of chunk c): K / 16 fragment blocks, then a 1-KiB
    vector page with fp32 v[32] and u[32] in the tile's accumulator order (index 16 hi + rho <-> channel 64 c + 32 hi + 16 u + rho);
    3 KiB of zero padding behind the last tile (the consumer fetches every (K / 16 + 1)-KiB tile as a 24-KiB stage)."""
    n, k = w.shape
    assert n % 64 == 0 and k % 64 == 0, (n, k)
    pi, kap = _rc_maps(k)
    rows = (64 * torch.arange(n // 64)[:, None, None] + pi[None]).to(w.device)          # [c, u, r]
    kap = kap.to(w.device)
    frag = w.detach()[rows[:, :, None, None, :, None], kap[None, None, :, :, None, :]].contiguous()   # [c, u, s, hi, r, j]
    fb = frag.reshape(n // 32, -1).view(torch.uint8)                                      # [tile, 64 K bytes]
    if not page:                       # K = 640 streams (tg_rc_linear): four 40-KiB tiles fill the LDS, the vectors travel as plain fp32 arrays
        return fb.contiguous().reshape(-1)
    page = torch.zeros(n // 32, 256, dtype=torch.float32, device=w.device)

    def tile_order(x):      # [N] -> [tile = 2 c + u, 16 hi + rho]
        return x.detach().float().reshape(n // 64, 2, 2, 16).permute(0, 2, 1, 3).reshape(n // 32, 32)
    if v is not None:
        page[:, :32] = tile_order(v)
    if u is not None:
        page[:, 32:64] = tile_order(u)
    out = torch.cat([fb, page.view(torch.uint8)], dim=1).contiguous().reshape(-1)
    return torch.cat([out, torch.zeros(3 * 1024, dtype=torch.uint8, device=w.device)])


def skinny_pack(w: torch.Tensor) -> torch.Tensor:
    """Linear weight [N, K] (N % 32 == 0, K % 16 == 0) -> tg_skinny_gemm's fragment order: block (nt, ks) = 64 lanes x 8 elements, lane (hi, l31), element j =
    W[32 nt + l31, 16 ks + 8 hi + j] — the A operand of one 32x32x16 MFMA as ONE contiguous KiB, so the weight stream is whole-line loads in K order."""
    n, k = w.shape
    assert n % 32 == 0 and k % 16 == 0, (n, k)
    return w.detach().reshape(n // 32, 32, k // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().reshape(-1)

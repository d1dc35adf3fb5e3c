import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops, rowchain
from dev_rc_linear import timeit
dev = "cuda"; dtype = torch.bfloat16
C, L = 320, 77
for (B, N, T) in [(16, 4096, 0), (16, 4096, 4), (2, 4096, 0)]:
    M = B * N
    h = torch.randn(M, C, device=dev).to(dtype)
    wq = (torch.randn(C, C, device=dev) / C ** 0.5).to(dtype); wo = wq.clone(); bo = torch.zeros(C, device=dev, dtype=dtype)
    g = torch.ones(C, device=dev, dtype=dtype); b_ = torch.zeros(C, device=dev, dtype=dtype)
    k = torch.randn(B * L, C, device=dev).to(dtype); vt = torch.randn(B, C, 80, device=dev).to(dtype)
    kip = torch.randn(B * max(T, 1), C, device=dev).to(dtype); vtip = torch.randn(B, C, 8 * ((max(T, 1) + 7) // 8), device=dev).to(dtype)
    wqp = rowchain.pack_xattn_q(wq, None, g, b_, 40 ** -0.5); wop = rowchain.pack_xattn_out(wo, bo)
    kv = ops.rc_kv_pack(k, vt, 80, L, kip if T else None, vtip if T else None, vtip.shape[2], T, B)
    out = torch.empty_like(h)
    row = {}
    for dbg in (0, 32 | (1 << 8), 32 | (2 << 8), 32 | (3 << 8), 32 | (4 << 8), 64 | (1 << 8), 64 | (2 << 8), 64 | (3 << 8), 64 | (4 << 8), 64 | (6 << 8)):
        row[dbg] = round(timeit(lambda i: ops.rc_xattn(h, wqp, kv, wop, N, 1e-5, T, out=out, text_len=77 | (dbg << 8))), 1)
    print(B, N, T, "dbg (1 no to_q, 2 no attn, 4 no to_out, 8 no dma, 16 no barrier):", row, flush=True)

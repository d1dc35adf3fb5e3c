"""dev helper (round 5): the fused norm2 + to_q + cross-attention launch (tg_xq_attn) vs the three-launch path's first two launches (LayerNorm-folded to_q GEMM +
flash attention) on the SD-1.5 bench's inner-level shapes (CFG batch 16); rotating activation sets, us"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
from theatergen_amd.weights_pack import pack_ln_linear
dev, dt = "cuda:0", torch.bfloat16
NC = 4
def timeit(fns, iters=24):
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (C, B, N, T) in [(640, 16, 1024, 4), (1280, 16, 256, 4), (640, 2, 2304, 4), (1280, 2, 1024, 16)]:
    heads, L = 8, 77
    d, M = C // heads, B * N
    xs = [torch.randn(M, C, device=dev).to(dt) for _ in range(NC)]
    wq = (torch.randn(C, C, device=dev) / C ** 0.5).to(dt)
    gamma, beta = torch.ones(C, device=dev).to(dt), torch.zeros(C, device=dev).to(dt)
    k = torch.randn(B * L, C, device=dev).to(dt); vt = torch.randn(B, C, 80, device=dev).to(dt)
    kip = torch.randn(B * T, C, device=dev).to(dt); vtip = torch.randn(B, C, 8 * ((T + 7) // 8), device=dev).to(dt)
    scale = d ** -0.5
    wl, u, v = pack_ln_linear(wq, None, gamma, beta)
    wx, ux, vx = pack_ln_linear(wq, None, gamma, beta, scale=scale * math.log2(math.e))
    blob = ops.xq_kv_pack(k, vt, 80, L, kip, vtip, vtip.shape[2], T, B, C, d)
    w1 = torch.full((1,), 0.4, device=dev)
    outs = [torch.empty(M, C, device=dev, dtype=dt) for _ in range(NC)]
    qs = [torch.empty(M, C, device=dev, dtype=dt) for _ in range(NC)]
    def old(i):
        ops.gemm(xs[i], wl, M, C, C, ln=(u, v, 1e-5), out=qs[i])
        ops.attention(qs[i], C, N * C, k, C, L * C, vt, 80, C * 80, L, B, heads, d, N, scale, outs[i], C, N * C, k1=kip, k1_ld=C, k1_bs=T * C, vt1=vtip,
                      vt1_ld=vtip.shape[2], vt1_bs=C * vtip.shape[2], len1=T, w1=0.4, w1_dev=w1)
    def new(i):
        ops.xq_attn(xs[i], wx, ux, vx, 1e-5, blob, d, N, L, T, ip_scale=w1, out=outs[i])
    fo = [(lambda i=i: old(i)) for i in range(NC)]; fn = [(lambda i=i: new(i)) for i in range(NC)]
    fq = [(lambda i=i: ops.gemm(xs[i], wl, M, C, C, ln=(u, v, 1e-5), out=qs[i])) for i in range(NC)]
    timeit(fo); to = timeit(fo); timeit(fn); tn = timeit(fn); timeit(fq); tq = timeit(fq)
    print(f"C={C} B={B} N={N} T={T}: to_q {tq:6.1f} us, to_q + attention {to:6.1f} us, fused {tn:6.1f} us", flush=True)

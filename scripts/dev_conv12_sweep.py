"""dev helper: the 12 x 12 level of the 768^2 plan (M = 288, weight-streaming): implicit-GEMM conv with forced tiles / K splits."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
dev, dt = "cuda:0", torch.bfloat16
def timeit(fns, iters=16):
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (B, h, cin, c1, cout) in [(2, 12, 1280, 0, 1280), (2, 12, 1280, 1280, 1280)]:
    g = torch.Generator().manual_seed(0)
    ctot = cin + c1; M = B * h * h
    X0 = torch.randn(M, cin, generator=g).to(dev, dt); X1 = torch.randn(M, c1, generator=g).to(dev, dt) if c1 else None
    Ws = [(torch.randn(cout, 9 * ctot, generator=g) / math.sqrt(9 * ctot)).to(dev, dt) for _ in range(4)]
    bias = torch.randn(cout, generator=g).to(dev, dt)
    out = []
    for ft, fs in [(0, 0), (1, 8), (1, 12), (1, 16), (1, 24), (4, 8), (4, 12), (4, 16), (4, 24), (2, 8), (2, 16), (3, 8), (3, 16)]:
        try:
            pl = ops.conv3x3(X0, Ws[0], B, h, h, cin, x1=X1, c1=c1, bias=bias, force_tile=ft, force_split_k=fs, plan_only=True)
            t = timeit([(lambda w=w: ops.conv3x3(X0, w, B, h, h, cin, x1=X1, c1=c1, bias=bias, force_tile=ft, force_split_k=fs)) for w in Ws])
            t = timeit([(lambda w=w: ops.conv3x3(X0, w, B, h, h, cin, x1=X1, c1=c1, bias=bias, force_tile=ft, force_split_k=fs)) for w in Ws])
            out.append(f"t{ft}s{fs}[{pl[0]}x{pl[1]}s{pl[2]}]:{t:6.1f}")
        except RuntimeError as e:
            out.append(f"t{ft}s{fs}:err")
    print(f"M={M} {ctot}->{cout}: " + " ".join(out), flush=True)

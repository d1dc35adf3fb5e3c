"""dev: where does the slab conv kernel's K-step period go?  TG_GEMM_FLAGS bit 9 = no weight DMA in the loop, bit 10 = no window loads"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
dev, dt = "cuda:0", torch.bfloat16
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (B, h, cin, cout) in [(16, 64, 320, 320), (16, 64, 960, 320), (16, 32, 640, 640), (16, 32, 1920, 640)]:
    g = torch.Generator().manual_seed(0)
    M = B * h * h
    x = torch.randn(M, cin, generator=g).to(dev, dt)
    W = (torch.randn(cout, 9 * cin, generator=g) / math.sqrt(9 * cin)).to(dev, dt)
    bias = torch.randn(cout, generator=g).to(dev, dt)
    out = torch.empty(M, cout, device=dev, dtype=dt)
    bvec = torch.randn(B, cout, generator=g).to(dev, dt)
    res = torch.randn(M, cout, generator=g).to(dev, dt)
    coef = torch.randn(B, 2, cin, generator=g).to(dev) if os.environ.get("SLAB_PRO") else None
    row = [f"{h}x{h} {cin}->{cout}"]
    for fl in (sys.argv[1].split() if len(sys.argv) > 1 else ("0", "512", "1024", "1536")):
        os.environ["TG_GEMM_FLAGS"] = fl
        us = timeit(lambda: ops.conv3x3(x, W, B, h, h, cin, bias=bias, bvec=bvec, rows_per_batch=h * h, res=res, out=out, a_coef=coef, a_silu=True))
        row.append(f"flags {fl}: {us:7.1f}us")
    tiles = M // 128 * (cout // 320)
    per = 9 * cin // 64 * 1280 * ((tiles + 255) // 256) / 2.2e3
    row.append(f"MFMA-only at 2.2 GHz: {per:6.1f}us")
    print("  ".join(row), flush=True)

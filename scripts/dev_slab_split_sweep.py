"""dev helper: the 3x3 convs the round-3 planner rules left to the halo / implicit-GEMM kernels (small-M levels of the batch-2 plans, the
8 x 8 level), timed on (a) the previous choice (TG_GEMM_FLAGS bit 11) and (b) the slab kernel with 1..8 K splits (force_tile 11 +
force_split_k), rotating operands; checks every slab result against (a).  Output feeds the cost constants of slab_splits_of (tg_gemm.hip)."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from theatergen_amd import ops

dev, dt = "cuda:0", torch.bfloat16
ROT = 3


def timeit(fns, iters=12):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def bench(B, h, w, cin, c1, cout):
    g = torch.Generator().manual_seed(0)
    ctot = cin + c1
    M = B * h * w
    X0 = [torch.randn(M, cin, generator=g).to(dev, dt) for _ in range(ROT)]
    X1 = [torch.randn(M, c1, generator=g).to(dev, dt) if c1 else None for _ in range(ROT)]
    W = (torch.randn(cout, 9 * ctot, generator=g) / math.sqrt(9 * ctot)).to(dev, dt)
    bias = torch.randn(cout, generator=g).to(dev, dt)
    bvec = torch.randn(B, cout, generator=g).to(dev, dt)
    res = torch.randn(M, cout, generator=g).to(dev, dt)
    O = [torch.empty(M, cout, device=dev, dtype=dt) for _ in range(ROT)]
    fl = 2.0 * M * cout * 9 * ctot
    kw = dict(bias=bias, bvec=bvec, rows_per_batch=h * w, res=res)

    def run(i, **extra):
        return ops.conv3x3(X0[i], W, B, h, w, cin, x1=X1[i], c1=c1, out=O[i], **kw, **extra)

    os.environ["TG_GEMM_FLAGS"] = "2048"
    kk = ops.conv3x3(X0[0], W, B, h, w, cin, x1=X1[0], c1=c1, plan_only=True, **kw)
    us0 = timeit([(lambda i=i: run(i)) for i in range(ROT)])
    us0 = timeit([(lambda i=i: run(i)) for i in range(ROT)])
    ref = O[0].float().clone()
    del os.environ["TG_GEMM_FLAGS"]
    kn = ops.conv3x3(X0[0], W, B, h, w, cin, x1=X1[0], c1=c1, plan_only=True, **kw)
    row = [f"B={B} {h}x{w} {cin}+{c1}->{cout} M={M} t={M // 128 * (cout // 320)} ch={ctot // 64}: prev kind{kk[3]} s{kk[2]} {us0:6.1f}us {fl / us0 / 1e6:4.0f}TF | new rule kind{kn[3]} s{kn[2]} | slab S:"]
    for S in (1, 2, 3, 4, 5, 6, 7, 8):
        if S > ctot // 64:
            break
        try:
            ops.conv3x3(X0[0], W, B, h, w, cin, x1=X1[0], c1=c1, plan_only=True, force_tile=11, force_split_k=S, **kw)
        except RuntimeError:
            continue
        us = timeit([(lambda i=i: run(i, force_tile=11, force_split_k=S)) for i in range(ROT)])
        err = ((O[0].float() - ref).norm() / ref.norm()).item()
        row.append(f" {S}:{us:6.1f}{'' if err < 6e-3 else f' ERR {err:.2e}'}")
    print("".join(row), flush=True)


if __name__ == "__main__":
    shapes = [
        # SD-2.1 768^2, CFG batch 2 (BASELINE configs[3])
        (2, 48, 48, 320, 0, 640), (2, 48, 48, 640, 0, 640), (2, 48, 48, 640, 320, 640), (2, 48, 48, 640, 640, 640), (2, 48, 48, 1280, 640, 640),
        (2, 24, 24, 640, 0, 1280), (2, 24, 24, 1280, 0, 1280), (2, 24, 24, 1280, 640, 1280), (2, 24, 24, 1280, 1280, 1280),
        # SDXL 1024^2, CFG batch 2 (configs[4])
        (2, 64, 64, 320, 0, 640), (2, 32, 32, 640, 0, 1280), (2, 32, 32, 1280, 0, 1280), (2, 32, 32, 1280, 640, 1280), (2, 32, 32, 1280, 1280, 1280),
        # SD-1.5 512^2, CFG batch 2 (configs[0]) and the 8 x 8 level at CFG batch 16 (configs[1])
        (2, 64, 64, 320, 0, 320), (2, 32, 32, 640, 0, 640), (2, 16, 16, 1280, 0, 1280), (2, 8, 8, 1280, 0, 1280),
        (16, 8, 8, 1280, 0, 1280), (16, 8, 8, 1280, 1280, 1280),
    ]
    for s in shapes:
        bench(*s)

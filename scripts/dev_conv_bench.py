"""dev helper: the SD-1.5 UNet's 3x3 convs at CFG batch 16, rotating operands: LDS-halo kernel (TG_GEMM_FLAGS=128) vs the slab
kernel, each alone and with its GroupNorm (old: tg_groupnorm + conv; new: tg_groupnorm_coef + conv with the prologue)."""
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


def bench(B, h, cin, c1, cout):
    g = torch.Generator().manual_seed(0)
    ctot = cin + c1
    M = B * h * h
    X0 = [torch.randn(M, cin, generator=g).to(dev, dt) for _ in range(ROT)]
    X1 = [torch.randn(M, c1, generator=g).to(dev, dt) if c1 else None for _ in range(ROT)]
    XC = [torch.cat([a, b], 1).contiguous() if b is not None else a for a, b in zip(X0, X1)]
    W = (torch.randn(cout, 9 * ctot, generator=g) / math.sqrt(9 * ctot)).to(dev, dt)
    bias = torch.randn(cout, generator=g).to(dev, dt)
    bvec = torch.randn(B, cout, generator=g).to(dev, dt)
    gam, bet = torch.ones(ctot, device=dev, dtype=dt), torch.zeros(ctot, device=dev, dtype=dt)
    O = [torch.empty(M, cout, device=dev, dtype=dt) for _ in range(ROT)]
    fl = 2.0 * M * cout * 9 * ctot
    row = [f"B={B} {h}x{h} {cin}+{c1}->{cout}"]
    kw = dict(bias=bias, bvec=bvec, rows_per_batch=h * h)
    os.environ["TG_GEMM_FLAGS"] = "128"
    kk = ops.conv3x3(XC[0], W, B, h, h, ctot, plan_only=True, **kw)
    us = timeit([(lambda i=i: ops.conv3x3(XC[i], W, B, h, h, ctot, out=O[i], **kw)) for i in range(ROT)])
    ref = O[0].clone()
    us_gn = timeit([(lambda i=i: ops.conv3x3(ops.groupnorm(X0[i], B, h * h, 32, 1e-5, gam, bet, silu=True, x1=X1[i]), W, B, h, h, ctot,
                                             out=O[i], **kw)) for i in range(ROT)])
    row.append(f"old kind{kk[3]} s{kk[2]}: {us:7.1f}us {fl / us / 1e6:5.0f}TF  gn+conv {us_gn:7.1f}us")
    del os.environ["TG_GEMM_FLAGS"]
    kk = ops.conv3x3(XC[0], W, B, h, h, ctot, plan_only=True, **kw)
    if kk[3] == 4:
        us2 = timeit([(lambda i=i: ops.conv3x3(XC[i], W, B, h, h, ctot, out=O[i], **kw)) for i in range(ROT)])
        same = torch.equal(O[0], ref)
        us2_gn = timeit([(lambda i=i: ops.conv3x3(X0[i], W, B, h, h, cin, x1=X1[i], c1=c1, out=O[i], a_silu=True,
                                                  a_coef=ops.groupnorm_coef(X0[i], B, h * h, 32, 1e-5, gam, bet, x1=X1[i]), **kw))
                         for i in range(ROT)])
        us_coef = timeit([(lambda i=i: ops.groupnorm_coef(X0[i], B, h * h, 32, 1e-5, gam, bet, x1=X1[i])) for i in range(ROT)])
        row.append(f"slab bm{kk[0]}: {us2:7.1f}us {fl / us2 / 1e6:5.0f}TF{'' if same else ' DIFF'}  coef+fused {us2_gn:7.1f}us (coef alone {us_coef:5.1f})")
    else:
        row.append(f"slab: not taken (kind {kk[3]})")
    print("  ".join(row), flush=True)


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    for case in [(64, 320, 0, 320), (64, 640, 320, 320), (64, 320, 320, 320), (32, 320, 0, 640), (32, 640, 0, 640), (32, 1280, 640, 640),
                 (32, 640, 640, 640), (32, 640, 320, 640), (16, 640, 0, 1280), (16, 1280, 0, 1280), (16, 1280, 1280, 1280), (16, 1280, 640, 1280)]:
        bench(B, *case)

"""dev timing (not part of the product): the slab conv on ping-pong compute waves (tg_conv_slab_pp.hip, default) against conv_slab_kernel (TG_SLAB_PP=0) on the
ResnetBlock2D convs of the SD-1.5 bench (CFG batch 16), GroupNorm + SiLU prologue on, isolated launches, rotating weights, random operands."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
from theatergen_amd.weights_pack import pack_conv3x3

dev, dt = "cuda:0", torch.bfloat16


def timeit(fns, iters=30):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for i in range(iters):
            fns[i % len(fns)]()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def with_env(k, v, fn):
    old = os.environ.get(k)
    os.environ[k] = v
    try:
        return fn()
    finally:
        if old is None:
            del os.environ[k]
        else:
            os.environ[k] = old


B = 16
print("layer (B = 16, + GroupNorm/SiLU prologue, bias + time vector + residual)      ping-pong (us, TF)    one wave / SIMD (us, TF)    launches per step")
tot_new = tot_old = 0.0
for (h, cin, c1, cout, n) in [(64, 320, 0, 320, 7), (64, 320, 320, 320, 2), (64, 640, 320, 320, 1), (32, 640, 0, 640, 6), (32, 320, 0, 640, 1), (32, 640, 320, 640, 1),
                              (32, 640, 640, 640, 1), (32, 1280, 640, 640, 1), (16, 1280, 0, 1280, 6), (16, 640, 0, 1280, 1), (16, 1280, 640, 1280, 1), (16, 1280, 1280, 1280, 2)]:
    g = torch.Generator().manual_seed(h + cin + c1 + cout)
    ctot = cin + c1
    M = B * h * h
    x0 = (torch.rand(M, cin, generator=g) * 2 - 1).to(dt).to(dev)
    x1 = (torch.rand(M, c1, generator=g) * 2 - 1).to(dt).to(dev) if c1 else None
    wps = [pack_conv3x3(((torch.rand(cout, ctot, 3, 3, generator=g) * 2 - 1) / math.sqrt(9 * ctot)).to(dt)).to(dev) for _ in range(3)]
    bias, bvec = torch.randn(cout, generator=g).to(dt).to(dev), torch.randn(B, cout, generator=g).to(dt).to(dev)
    res = torch.randn(M, cout, generator=g).to(dt).to(dev)
    gamma, beta = torch.ones(ctot, device=dev, dtype=dt), torch.zeros(ctot, device=dev, dtype=dt)
    coef = ops.groupnorm_coef(x0, B, h * h, 32, 1e-5, gamma, beta, x1=x1)
    fns = [(lambda wp=wp: ops.conv3x3(x0, wp, B, h, h, cin, x1=x1, c1=c1, bias=bias, bvec=bvec, rows_per_batch=h * h, res=res, a_coef=coef, a_silu=True)) for wp in wps]
    fl = 2.0 * M * cout * 9 * ctot
    t_new = timeit(fns)
    t_old = with_env("TG_SLAB_PP", "0", lambda: timeit(fns))
    a = fns[0]().float()
    b = with_env("TG_SLAB_PP", "0", lambda: fns[0]().float())
    err = ((a - b).norm() / b.norm()).item()
    tot_new += n * t_new
    tot_old += n * t_old
    print(f"{h:3d}x{h:<3d} {ctot:5d} -> {cout:5d}    {t_new:7.1f} {fl / t_new / 1e6:5.0f}      {t_old:7.1f} {fl / t_old / 1e6:5.0f}      x{n}   rel diff {err:.1e}", flush=True)
print(f"per step (launch-weighted): ping-pong {tot_new:.0f} us, one wave / SIMD {tot_old:.0f} us")


# ---- where does a K-step go?  dev switches of the two-waves-per-SIMD kernel (wrong results by design)
print("dev switches (us):")
for (h, cin, c1, cout) in [(64, 640, 320, 320), (32, 640, 0, 640), (16, 1280, 0, 1280)]:
    g = torch.Generator().manual_seed(1)
    ctot, M = cin + c1, B * h * h
    x0 = (torch.rand(M, cin, generator=g) * 2 - 1).to(dt).to(dev)
    x1 = (torch.rand(M, c1, generator=g) * 2 - 1).to(dt).to(dev) if c1 else None
    wps = [pack_conv3x3(((torch.rand(cout, ctot, 3, 3, generator=g) * 2 - 1) / math.sqrt(9 * ctot)).to(dt)).to(dev) for _ in range(3)]
    bias = torch.randn(cout, generator=g).to(dt).to(dev)
    coef = ops.groupnorm_coef(x0, B, h * h, 32, 1e-5, torch.ones(ctot, device=dev, dtype=dt), torch.zeros(ctot, device=dev, dtype=dt), x1=x1)
    fns = [(lambda wp=wp: ops.conv3x3(x0, wp, B, h, h, cin, x1=x1, c1=c1, bias=bias, a_coef=coef, a_silu=True)) for wp in wps]
    row = f"{h}x{h} {ctot}->{cout}: "
    for name, fl in (("full", 0), ("no weight DMA", 1 << 16), ("no window staging", 1 << 17), ("no DMA, no window", 3 << 16), ("no MFMA", 1 << 18),
                     ("loaders only (no MFMA)", 1 << 18), ("no DMA, no window, no MFMA", 7 << 16), ("old kernel", -1)):
        if fl < 0:
            t = with_env("TG_SLAB_PP", "0", lambda: timeit(fns))
        else:
            t = with_env("TG_GEMM_FLAGS", str(fl), lambda: timeit(fns))
        row += f"{name} {t:.1f} | "
    print(row, flush=True)

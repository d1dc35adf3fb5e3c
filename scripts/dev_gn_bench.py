"""dev helper (not part of the product or the tests): GroupNorm (+SiLU) time per UNet shape at CFG batch 16 against the
bytes it has to move (read twice, write once) and a plain device copy of the same tensor.  Usage: python scripts/dev_gn_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from theatergen_amd import ops

dev, dt, B = "cuda:0", torch.bfloat16, 16


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for hw, c0, c1 in [(4096, 320, 0), (4096, 640, 320), (4096, 320, 320), (1024, 640, 0), (1024, 1280, 640), (256, 1280, 0),
                   (256, 1280, 1280), (64, 1280, 0), (64, 1280, 1280)]:
    C = c0 + c1
    x0 = torch.randn(B * hw, c0, device=dev).to(dt)
    x1 = torch.randn(B * hw, c1, device=dev).to(dt) if c1 else None
    gamma = torch.randn(C, device=dev).to(dt)
    beta = torch.randn(C, device=dev).to(dt)
    out = torch.empty(B * hw, C, device=dev, dtype=dt)
    us = timeit(lambda: ops.groupnorm(x0, B, hw, 32, 1e-5, gamma, beta, silu=True, x1=x1, out=out))
    big = torch.empty(B * hw, C, device=dev, dtype=dt)
    us_copy = timeit(lambda: out.copy_(big))
    mb = B * hw * C * 2 / 1e6
    print(f"hw={hw:5d} C={c0:4d}+{c1:4d} ({mb:6.1f} MB): groupnorm {us:7.1f} us = {3 * mb / us:5.2f} TB/s (2R+1W)   "
          f"copy {us_copy:7.1f} us = {2 * mb / us_copy:5.2f} TB/s", flush=True)

"""dev helper: big-tile kernels (force_tile 9 = 128 x 320, 10 = 256 x 256) against the 128 x 128 kernel (1) and the heuristic (0)
on the SD-1.5 UNet's plain-GEMM shapes at CFG batch 16, operands rotated over several copies (not L2-warm), bias + residual
in the epilogue where the layer has them, fused GEGLU for FF1."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from theatergen_amd import ops
from theatergen_amd.weights_pack import pack_geglu

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


def bench(name, M, N, K, res=False, geglu=False, tiles=(0, 1, 9, 10)):
    g = torch.Generator().manual_seed(0)
    A = [torch.randn(M, K, generator=g).to(dev, dt) for _ in range(ROT)]
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    if geglu:
        Wp, bp = pack_geglu(W.to(dt), b.to(dt))
        Wd, bd = Wp.to(dev), bp.to(dev)
    else:
        Wd, bd = W.to(dev, dt), b.to(dev, dt)
    R = [torch.randn(M, N, generator=g).to(dev, dt) for _ in range(ROT)] if res else [None] * ROT
    O = [torch.empty(M, N // 2 if geglu else N, device=dev, dtype=dt) for _ in range(ROT)]
    row = [f"{name:10s} M={M:6d} N={N:5d} K={K:5d}{' +res' if res else '     '}{' geglu' if geglu else '      '}"]
    base = None
    for t in tiles:
        if geglu and t == 9:
            continue
        try:
            fns = [(lambda i=i: ops.gemm(A[i], Wd, M, N, K, bias=bd, res=R[i], geglu=geglu, out=O[i], force_tile=t)) for i in range(ROT)]
            us = timeit(fns)
            out = O[0].clone()
            if base is None:
                base = out
            ok = torch.equal(out, base)
            row.append(f"t{t}: {us:7.1f}us {2.0 * M * N * K / us / 1e6:6.0f}TF{'' if ok else ' DIFF'}")
        except RuntimeError as e:
            row.append(f"t{t}: fail {str(e)[-40:]}")
    print("  ".join(row), flush=True)


if __name__ == "__main__":
    bench("L0 proj", 65536, 320, 320, res=True)
    bench("L0 to_q", 65536, 320, 320)
    bench("L0 qkv", 65536, 960, 320)
    bench("L0 ff1", 65536, 2560, 320, geglu=True)
    bench("L0 ff2", 65536, 320, 1280, res=True)
    bench("L1 proj", 16384, 640, 640, res=True)
    bench("L1 qkv", 16384, 1920, 640)
    bench("L1 ff1", 16384, 5120, 640, geglu=True)
    bench("L1 ff2", 16384, 640, 2560, res=True)
    bench("L2 proj", 4096, 1280, 1280, res=True)
    bench("L2 qkv", 4096, 3840, 1280)
    bench("L2 ff1", 4096, 10240, 1280, geglu=True)
    bench("L2 ff2", 4096, 1280, 5120, res=True)
    bench("L3 proj", 1024, 1280, 1280, res=True)
    bench("big", 8192, 4096, 4096)
    bench("big2", 16384, 5120, 2560)

"""dev helper: loader/compute GEMM (heuristic) (TG_GEMM_FLAGS=256) vs the 128x128 kernel (0) on the SD-1.5 UNet's long-K plain GEMMs at
CFG batch 16, rotating operands, bias + residual epilogue; each variant timed twice (first pass of a shape is the warm-up)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
dev, dt = "cuda:0", torch.bfloat16
ROT = 3
def timeit(fns, iters=15):
    for f in fns: f()
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (name, M, N, K) in [("ff2 64", 65536, 320, 1280), ("ff2 32", 16384, 640, 2560), ("ff2 16", 4096, 1280, 5120), ("proj 16", 4096, 1280, 1280),
                        ("ff2 8", 1024, 1280, 5120), ("big", 8192, 3840, 4096)]:
    g = torch.Generator().manual_seed(0)
    A = [torch.randn(M, K, generator=g).to(dev, dt) for _ in range(ROT)]
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev, dt)
    b = torch.randn(N, generator=g).to(dev, dt)
    R = [torch.randn(M, N, generator=g).to(dev, dt) for _ in range(ROT)]
    O = [torch.empty(M, N, device=dev, dtype=dt) for _ in range(ROT)]
    row = [f"{name:8s} M={M:6d} N={N:5d} K={K:5d}"]
    for rep in range(2):
        for fl in ("256", "0"):
            os.environ["TG_GEMM_FLAGS"] = fl
            kk = ops.gemm(A[0], W, M, N, K, bias=b, res=R[0], plan_only=True)
            us = timeit([(lambda i=i: ops.gemm(A[i], W, M, N, K, bias=b, res=R[i], out=O[i])) for i in range(ROT)])
            if rep == 1:
                row.append(f"flags {fl} kind{kk[3]} {kk[0]}x{kk[1]} s{kk[2]}: {us:7.1f}us {2.0 * M * N * K / us / 1e6:5.0f}TF")
    print("  ".join(row), flush=True)

"""dev helper: does operand DATA (zeros vs random) change GEMM speed?  (power / clock effect)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
dev, dt = "cuda:0", torch.bfloat16
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (M, N, K) in [(8192, 4096, 4096), (65536, 320, 320), (65536, 2560, 320), (16384, 640, 2560), (2048, 2048, 1280)]:
    for kind in ["zeros", "randn", "small-int"]:
        if kind == "zeros":
            a = torch.zeros(M, K, device=dev, dtype=dt); w = torch.zeros(N, K, device=dev, dtype=dt)
        elif kind == "randn":
            a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
        else:
            a = torch.ones(M, K, device=dev, dtype=dt); w = torch.ones(N, K, device=dev, dtype=dt)
        ms = timeit(lambda: ops.linear(a, w))
        print(f"M={M} N={N} K={K} {kind:10s} {2.0 * M * N * K / ms / 1e9:7.0f} TF  {ms * 1e3:8.1f} us", flush=True)

"""dev helper: fixed per-tile cost vs per-K-tile cost of the plain GEMM (K sweep at fixed M, N)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ACT = int(os.environ.get('ACT', '0'))
FT = int(os.environ.get('FT', '0'))
from theatergen_amd import ops
dev, dt = "cuda:0", torch.bfloat16
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (M, N) in [(65536, 2560), (65536, 320)]:
    tiles = (M // 128) * ((N + 127) // 128)
    for K in [64, 320, 1280]:
        a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt); b = torch.randn(N, device=dev).to(dt)
        ms = timeit(lambda: ops.linear(a, w, b, act=ACT, force_tile=FT))
        rounds = tiles / 512
        print(f"M={M} N={N} K={K:5d} tiles={tiles} {2.0 * M * N * K / ms / 1e9:7.0f} TF {ms * 1e3:8.1f} us  per-tile-round {ms * 1e3 / max(rounds, 1):6.2f} us  ({K // 64} K-tiles)", flush=True)

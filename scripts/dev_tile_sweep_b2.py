"""dev helper: tile / split choice for the plain GEMMs of the CFG-batch-2 plans (BASELINE configs[3] / configs[4]): every forced tile id,
forced K splits on the default tile, the big-tile and loader / compute kernels; rotating weight copies, us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
dev, dt = "cuda:0", torch.bfloat16
NCOPY = 4
def timeit(fns, iters=16):
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def case(M, N, K, geglu=False):
    a = torch.randn(M, K, device=dev).to(dt); ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(dt) for _ in range(NCOPY)]
    b = torch.randn(N, device=dev).to(dt); r = None if geglu else torch.randn(M, N, device=dev).to(dt)
    out = []
    for ft, fs in [(0, 0), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (6, 0), (7, 0), (9, 0), (10, 0), (13, 0), (14, 0), (1, 2), (1, 3), (1, 4), (2, 2), (5, 2)]:
        try:
            pl = ops.gemm(a, ws[0], M, N, K, bias=b, res=r, geglu=geglu, force_tile=ft, force_split_k=fs, plan_only=True)
            t = timeit([(lambda w=w: ops.gemm(a, w, M, N, K, bias=b, res=r, geglu=geglu, force_tile=ft, force_split_k=fs)) for w in ws])
            t = timeit([(lambda w=w: ops.gemm(a, w, M, N, K, bias=b, res=r, geglu=geglu, force_tile=ft, force_split_k=fs)) for w in ws])
            out.append(f"t{ft}" + (f"s{fs}" if fs else "") + f"[{pl[0]}x{pl[1]}k{pl[3]}s{pl[2]}]:{t:6.1f}")
        except RuntimeError as e:
            pass
    print(f"M={M:6d} N={N:5d} K={K:5d} {'geglu' if geglu else '     '} {2.0 * M * N * K / 1e9:6.1f} GF  " + " ".join(out), flush=True)
# SDXL level 2 (32 x 32 x batch 2), level 1 (64 x 64 x 2); SD-2.1 levels 0..3 (96 / 48 / 24 / 12 squared x 2)
for (M, c) in [(2048, 1280), (8192, 640), (18432, 320), (4608, 640), (1152, 1280), (288, 1280)]:
    case(M, c, c); case(M, 3 * c, c); case(M, c, 4 * c); case(M, 8 * c, c, geglu=True)

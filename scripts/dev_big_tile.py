import os, sys
sys.path.insert(0, "/root/repo")
import torch
from theatergen_amd import ops
dev, dt = "cuda:0", torch.bfloat16
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (M, N, K, G) in [(65536, 2560, 320, True), (16384, 5120, 640, True), (4096, 10240, 1280, True), (1024, 10240, 1280, True),
                     (65536, 2560, 320, False), (4096, 3840, 1280, False), (65536, 1280, 320, False), (16384, 2560, 640, False)]:
    a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt); b = torch.randn(N, device=dev).to(dt)
    ref = None
    row = []
    for ft in (0, 6):
        out = ops.gemm(a, w, M, N, K, bias=b, geglu=G, force_tile=ft)
        if ref is None: ref = out
        err = float((out.float() - ref.float()).abs().max())
        ms = timeit(lambda: ops.gemm(a, w, M, N, K, bias=b, geglu=G, force_tile=ft))
        row.append(f"ft{ft}: {2.0 * M * N * K / ms / 1e9:6.0f} TF (diff {err:.3g})")
    print(M, N, K, "geglu" if G else "", "  ".join(row), flush=True)

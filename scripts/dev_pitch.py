"""dev (round 5): does the K pitch of the operands (a power-of-two multiple: rows of a K-tile fall on few L2 channels) bound the GEMM K loop?
Same kernel, K vs K + 64 (row pitch an odd multiple of 128 bytes), TFLOP/s per case; rotating operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
dev, dt = "cuda:0", torch.bfloat16
NC = 4
def timeit(fns, iters=24):
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K0) in [(4096, 1280, 5120), (16384, 640, 2560), (4096, 1280, 1280), (16384, 640, 640), (4096, 10240, 1280), (8192, 4096, 4096), (65536, 320, 1280)]:
    row = []
    for K in (K0, K0 + 64, K0 + 192):
        As = [torch.randn(M, K, device=dev).to(dt) for _ in range(NC)]
        ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(dt) for _ in range(NC)]
        outs = [torch.empty(M, N, device=dev, dtype=dt) for _ in range(NC)]
        for ft in (0, 1):
            fns = [(lambda i=i: ops.gemm(As[i], ws[i], M, N, K, out=outs[i], force_tile=ft)) for i in range(NC)]
            timeit(fns); t = timeit(fns)
            pl = ops.gemm(As[0], ws[0], M, N, K, force_tile=ft, plan_only=True)
            row.append(f"K={K} ft{ft}[{pl[0]}x{pl[1]}]: {t:6.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TF")
        del As, ws, outs
    print(f"M={M} N={N}: " + " | ".join(row), flush=True)

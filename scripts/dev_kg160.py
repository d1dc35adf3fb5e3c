"""dev helper (round 5): the K-group tiles (64 x 160 / 32 x 160: force_tile 24 / 25) against the planner's choice on the M = 1024 / 2048 projection shapes
(SD-1.5 8 x 8 level; SDXL / SD-2.1 batch-2 plans); rotating operand sets, us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
dev = "cuda:0"
NCOPY = 4
def timeit(fns, iters=40):
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def case(M, N, K, res=True, dt=torch.bfloat16):
    As = [torch.randn(M, K, device=dev).to(dt) for _ in range(NCOPY)]
    ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(dt) for _ in range(NCOPY)]
    b = torch.randn(N, device=dev).to(dt)
    rs = [torch.randn(M, N, device=dev).to(dt) for _ in range(NCOPY)] if res else [None] * NCOPY
    outs = [torch.empty(M, N, device=dev, dtype=dt) for _ in range(NCOPY)]
    ref = ops.gemm(As[0], ws[0], M, N, K, bias=b, res=rs[0], force_tile=1).float()
    row = []
    for ft in (0, 2, 3, 21, 24, 25):
        try:
            pl = ops.gemm(As[0], ws[0], M, N, K, bias=b, res=rs[0], force_tile=ft, plan_only=True)
            got = ops.gemm(As[0], ws[0], M, N, K, bias=b, res=rs[0], force_tile=ft).float()
            err = ((got - ref).norm() / ref.norm()).item()
            fns = [(lambda i=i: ops.gemm(As[i], ws[i], M, N, K, bias=b, res=rs[i], out=outs[i], force_tile=ft)) for i in range(NCOPY)]
            timeit(fns)
            t = timeit(fns)
            row.append(f"ft{ft}[{pl[0]}x{pl[1]}s{pl[2]}]{'' if err < 3e-3 else '!DIFF%.1e' % err}:{t:6.1f}")
        except RuntimeError as e:
            row.append(f"ft{ft}:ERR {str(e)[:40]}")
    print(f"M={M:6d} N={N:5d} K={K:5d} res={int(res)} {2.0 * M * N * K / 1e9:6.1f} GF  " + " ".join(row), flush=True)
for (M, N, K, res) in [(1024, 1280, 1280, True), (1024, 1280, 1280, False), (1024, 1280, 2560, False), (1024, 1280, 5120, True), (1024, 1280, 640, False),
                       (2048, 1280, 1280, True), (2048, 1280, 1280, False), (2048, 1280, 5120, True), (2048, 1280, 2560, False), (2048, 1280, 640, False),
                       (2048, 640, 640, True), (2048, 640, 2560, True), (512, 1280, 1280, True), (512, 1280, 5120, True),
                       (4096, 640, 640, True), (4096, 640, 2560, True), (4096, 1280, 1280, True), (3072, 1280, 1280, True)]:
    case(M, N, K, res)

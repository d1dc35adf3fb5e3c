"""dev helper (round 5): the 128 x 160 tile variants (force_tile 21 / 22 / 23) against the planner's 128 x 128 choice on the SD-1.5 bench's mid-level
projection shapes; rotating operand sets (the in-situ launches read what another kernel just wrote), us per launch, bit-identity check."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TG_T160"] = "0"
import torch
from theatergen_amd import ops
dev, dt = "cuda:0", torch.bfloat16
NCOPY = 4
def timeit(fns, iters=24):
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def case(M, N, K, res=True):
    As = [torch.randn(M, K, device=dev).to(dt) for _ in range(NCOPY)]
    ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(dt) for _ in range(NCOPY)]
    b = torch.randn(N, device=dev).to(dt)
    rs = [torch.randn(M, N, device=dev).to(dt) for _ in range(NCOPY)] if res else [None] * NCOPY
    outs = [torch.empty(M, N, device=dev, dtype=dt) for _ in range(NCOPY)]
    ref = ops.gemm(As[0], ws[0], M, N, K, bias=b, res=rs[0], force_tile=1).clone()
    row = []
    for ft in (0, 1, 7, 21, 23):
        try:
            pl = ops.gemm(As[0], ws[0], M, N, K, bias=b, res=rs[0], force_tile=ft, plan_only=True)
            got = ops.gemm(As[0], ws[0], M, N, K, bias=b, res=rs[0], force_tile=ft)
            same = torch.equal(got, ref)
            fns = [(lambda i=i: ops.gemm(As[i], ws[i], M, N, K, bias=b, res=rs[i], out=outs[i], force_tile=ft)) for i in range(NCOPY)]
            timeit(fns)
            t = timeit(fns)
            row.append(f"ft{ft}[{pl[0]}x{pl[1]}s{pl[2]}]{'' if same else '!DIFF'}:{t:6.1f}")
        except RuntimeError as e:
            row.append(f"ft{ft}:ERR {str(e)[:40]}")
    print(f"M={M:6d} N={N:5d} K={K:5d} res={int(res)} {2.0 * M * N * K / 1e9:6.1f} GF  " + " ".join(row), flush=True)
for (M, N, K, res) in [(16384, 640, 640, True), (16384, 640, 640, False), (16384, 640, 2560, True), (16384, 640, 1280, False), (16384, 640, 1920, False), (16384, 640, 960, False),
                       (4096, 1280, 1280, True), (4096, 1280, 1280, False), (4096, 1280, 5120, True), (4096, 1280, 2560, False), (4096, 1280, 1920, False), (4096, 1280, 640, False),
                       (65536, 320, 1280, True), (65536, 320, 640, False), (65536, 320, 960, False), (65536, 320, 320, True),
                       (2048, 1280, 1280, True), (8192, 640, 640, True), (8192, 640, 2560, True), (2048, 1280, 5120, True)]:
    case(M, N, K, res)

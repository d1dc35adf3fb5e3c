"""dev helper: tile choice per UNet GEMM shape class (rotating weight copies)."""
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
    for ft in (0, 1, 2, 3, 4, 5, 6):
        if geglu and ft not in (0, 1, 2, 5, 6): continue
        try:
            t = timeit([(lambda w=w: ops.gemm(a, w, M, N, K, bias=b, res=r, geglu=geglu, force_tile=ft)) for w in ws])
            out.append(f"t{ft}:{t:7.1f}")
        except RuntimeError as e:
            out.append(f"t{ft}: err")
    print(f"M={M:6d} N={N:5d} K={K:5d} {'geglu' if geglu else '     '}  " + " ".join(out), flush=True)
for (M, c) in [(65536, 320), (16384, 640), (4096, 1280), (1024, 1280)]:
    case(M, c, c); case(M, 3 * c, c); case(M, c, 4 * c); case(M, 8 * c, c, geglu=True)

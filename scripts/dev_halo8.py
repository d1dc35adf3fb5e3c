"""dev helper (round 5): the 8 x 8 level's conv3x3 at CFG batch 16 (M = 1024) — LDS-halo kernel on 128 x 128 tiles (TG_T160=7) vs 128 x 160 tiles, one workgroup
per CU (TG_T160=15); rotating weight / activation sets, us per launch incl. the split reduce"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
from theatergen_amd.weights_pack import pack_conv3x3
dev, dt = "cuda:0", torch.bfloat16
NC = int(os.environ.get('NC', '4'))
def timeit(fns, iters=24):
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (B, cin, c1, cout) in [(16, 1280, 0, 1280), (16, 1280, 1280, 1280), (16, 1280, 640, 1280)]:
    M = B * 64
    xs = [torch.randn(M, cin, device=dev).to(dt) for _ in range(NC)]
    x1s = [torch.randn(M, c1, device=dev).to(dt) if c1 else None for _ in range(NC)]
    ws = [pack_conv3x3((torch.randn(cout, cin + c1, 3, 3, device=dev) / (9 * (cin + c1)) ** 0.5).to(dt)) for _ in range(NC)]
    b = torch.randn(cout, device=dev).to(dt); bv = torch.randn(B, cout, device=dev).to(dt)
    row = []
    for mode in ("7", "15"):
        os.environ["TG_T160"] = mode
        pl = ops.conv3x3(xs[0], ws[0], B, 8, 8, cin, x1=x1s[0], c1=c1, bias=b, bvec=bv, rows_per_batch=64, plan_only=True)
        fns = [(lambda i=i: ops.conv3x3(xs[i], ws[i], B, 8, 8, cin, x1=x1s[i], c1=c1, bias=b, bvec=bv, rows_per_batch=64)) for i in range(NC)]
        timeit(fns); t = timeit(fns)
        row.append(f"T160={mode} plan {pl}: {t:6.1f} us")
    print(f"8x8 conv B={B} cin={cin}+{c1} cout={cout} {2.0 * M * cout * 9 * (cin + c1) / 1e9:6.1f} GF  " + "  ".join(row), flush=True)

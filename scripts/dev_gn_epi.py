"""dev helper (not part of the product): cost of the GroupNorm partial sums in the two-wave slab conv's epilogue and of the statistics launches they replace."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
from theatergen_amd.weights_pack import pack_conv3x3

dev, dt = "cuda:0", torch.bfloat16


def timeit(f, iters=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


for (B, h, w, cin, cout) in [(16, 64, 64, 320, 320), (16, 64, 64, 640, 320), (16, 32, 32, 640, 640), (16, 32, 32, 1280, 640)]:
    x = torch.randn(B * h * w, cin, device=dev).to(dt)
    wp = pack_conv3x3((torch.randn(cout, cin, 3, 3) / math.sqrt(9 * cin)).to(dt)).to(dev)
    bias = torch.randn(cout, device=dev).to(dt)
    res = torch.randn(B * h * w, cout, device=dev).to(dt)
    gamma, beta = torch.ones(cout, device=dev).to(dt), torch.zeros(cout, device=dev).to(dt)
    out = torch.empty(B * h * w, cout, device=dev, dtype=dt)
    t0 = timeit(lambda: ops.conv3x3(x, wp, B, h, w, cin, bias=bias, res=res, out=out))
    gn = {"groups": 32}
    t1 = timeit(lambda: ops.conv3x3(x, wp, B, h, w, cin, bias=bias, res=res, out=out, gn_out=gn))
    t2 = timeit(lambda: ops.groupnorm_coef(out, B, h * w, 32, 1e-5, gamma, beta))
    t3 = timeit(lambda: ops.groupnorm_from_partials(gn, B, h * w, cout, 1e-5, gamma, beta))
    print(f"{B}x{h}x{w} {cin}->{cout}: conv {t0:7.1f} us, + partials {t1:7.1f} us ({t1 - t0:+.1f}); groupnorm_coef {t2:6.1f} us, from partials {t3:6.1f} us", flush=True)

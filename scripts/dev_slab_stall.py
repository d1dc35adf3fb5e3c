import os, sys
sys.path.insert(0, "/root/repo")
import torch
from theatergen_amd import ops
from theatergen_amd.weights_pack import pack_conv3x3
dev, dt = "cuda:0", torch.bfloat16
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (h, cin, cout) in [(64, 320, 320), (32, 640, 640), (16, 1280, 1280), (8, 1280, 1280)]:
    B = 16
    x = torch.randn(B * h * h, cin, device=dev).to(dt)
    ws = [pack_conv3x3((torch.randn(cout, cin, 3, 3, device=dev) / (9 * cin) ** 0.5).to(dt)) for _ in range(4)]
    b = torch.randn(cout, device=dev).to(dt)
    for sc in (1.0, 1.000001):
        fns = [(lambda w=w: ops.conv3x3(x, w, B, h, h, cin, bias=b, out_scale=sc)) for w in ws]
        for f in fns: f()
        t = sum(timeit(f, 5) for f in fns) / len(fns)
        print(f"{h}x{h} {cin}->{cout} out_scale={sc}: {t:7.1f} us", flush=True)

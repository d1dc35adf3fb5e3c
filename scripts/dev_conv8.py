"""dev helper: the 8x8 weight-streaming convs (M = 1024) under different tiles / splits, rotating weight copies."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
from theatergen_amd.weights_pack import pack_conv3x3
dev, dt = "cuda:0", torch.bfloat16
NCOPY = 6
def timeit(fns, iters=18):
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def conv_case(B, h, cin, cout, c1=0, stride=1):
    ctot = cin + c1
    x0 = torch.randn(B * h * h, cin, device=dev).to(dt); x1 = torch.randn(B * h * h, c1, device=dev).to(dt) if c1 else None
    ws = [pack_conv3x3((torch.randn(cout, ctot, 3, 3, device=dev) / (9 * ctot) ** 0.5).to(dt)) for _ in range(NCOPY)]
    b = torch.randn(cout, device=dev).to(dt)
    out = []
    for ft in (0, 3, 4, 5):
        for fs in (0, 2, 3, 4, 6, 8):
            try:
                t = timeit([(lambda w=w: ops.conv3x3(x0, w, B, h, h, cin, x1=x1, c1=c1, stride=stride, bias=b, force_split_k=fs, force_tile=ft)) for w in ws])
                out.append(f"t{ft}s{fs or 'a'}:{t:6.1f}")
            except RuntimeError as e:
                out.append(f"t{ft}s{fs}: err")
    print(f"conv {h}x{h} s{stride} {cin}+{c1}->{cout}  " + " ".join(out), flush=True)
conv_case(16, 8, 1280, 1280); conv_case(16, 8, 1280, 1280, 1280); conv_case(16, 16, 1280, 1280, stride=2); conv_case(16, 32, 640, 640, stride=2); conv_case(16, 64, 320, 320, stride=2)

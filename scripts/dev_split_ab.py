"""dev helper: tail-split plan vs no split, per shape, with rotating weight copies (cold-ish weights as in situ)."""
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
def gemm_case(M, N, K):
    a = torch.randn(M, K, device=dev).to(dt); ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(dt) for _ in range(NCOPY)]
    b = torch.randn(N, device=dev).to(dt); r = torch.randn(M, N, device=dev).to(dt)
    out = []
    for fs in (0, 1, 2, 3, 4, 6):
        try:
            out.append(f"s={fs or 'auto'}:{timeit([(lambda w=w: ops.linear(a, w, b, res=r, force_split_k=fs)) for w in ws]):7.1f}us")
        except RuntimeError as e:
            out.append(f"s={fs}: err")
    print(f"gemm M={M:6d} N={N:5d} K={K:5d}  " + "  ".join(out), flush=True)
def conv_case(B, h, cin, cout, c1=0):
    ctot = cin + c1
    x0 = torch.randn(B * h * h, cin, device=dev).to(dt); x1 = torch.randn(B * h * h, c1, device=dev).to(dt) if c1 else None
    ws = [pack_conv3x3((torch.randn(cout, ctot, 3, 3, device=dev) / (9 * ctot) ** 0.5).to(dt)) for _ in range(NCOPY)]
    b = torch.randn(cout, device=dev).to(dt)
    out = []
    for fs in (0, 1):
        out.append(f"s={fs or 'auto'}:{timeit([(lambda w=w: ops.conv3x3(x0, w, B, h, h, cin, x1=x1, c1=c1, bias=b, force_split_k=fs)) for w in ws]):7.1f}us")
    print(f"conv {h}x{h} {cin}+{c1}->{cout}  " + "  ".join(out), flush=True)
gemm_case(16384, 640, 2560); gemm_case(16384, 640, 640); gemm_case(4096, 1280, 5120); gemm_case(4096, 1280, 1280)
gemm_case(1024, 1280, 1280); gemm_case(1024, 3840, 1280); gemm_case(1024, 10240, 1280); gemm_case(1024, 1280, 5120)
conv_case(16, 32, 640, 640); conv_case(16, 32, 1280, 640, 0); conv_case(16, 16, 1280, 1280); conv_case(16, 16, 1280, 1280, 1280)
conv_case(16, 8, 1280, 1280); conv_case(16, 8, 1280, 1280, 1280)

"""dev reference point (not used by the product): the vendor GEMM (torch.addmm -> hipBLASLt / rocBLAS) and MIOpen conv on the
UNet's shapes next to tg_gemm / the halo conv, same bf16 operands, rotating weight copies."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from theatergen_amd import ops
from theatergen_amd.weights_pack import pack_conv3x3
dev, dt = "cuda:0", torch.bfloat16
NC = 4
def timeit(fns, iters=16):
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
print("GEMM out = x @ W^T + bias + res      ours(us)  vendor addmm+add(us)  vendor matmul only(us)")
for (M, N, K) in [(65536, 320, 320), (65536, 960, 320), (65536, 320, 1280), (65536, 2560, 320), (16384, 640, 640), (16384, 640, 2560),
                  (16384, 5120, 640), (4096, 1280, 1280), (4096, 1280, 5120), (4096, 10240, 1280), (8192, 4096, 4096)]:
    a = torch.randn(M, K, device=dev).to(dt); ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(dt) for _ in range(NC)]
    b = torch.randn(N, device=dev).to(dt); r = torch.randn(M, N, device=dev).to(dt)
    ours = timeit([(lambda w=w: ops.linear(a, w, b, res=r)) for w in ws])
    lib = timeit([(lambda w=w: torch.addmm(b, a, w.t()).add_(r)) for w in ws])
    mm = timeit([(lambda w=w: torch.matmul(a, w.t())) for w in ws])
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:5d} K={K:5d}   {ours:8.1f} ({fl / ours / 1e6:5.0f} TF)   {lib:8.1f} ({fl / lib / 1e6:5.0f} TF)   {mm:8.1f} ({fl / mm / 1e6:5.0f} TF)", flush=True)
print("conv3x3 (batch 16)                    ours(us)  MIOpen NCHW channels_last(us)")
for (h, cin, cout) in [(64, 320, 320), (32, 640, 640), (16, 1280, 1280), (8, 1280, 1280)]:
    B = 16
    x = torch.randn(B, cin, h, h, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    wt = [(torch.randn(cout, cin, 3, 3, device=dev) / (9 * cin) ** 0.5).to(dt).contiguous(memory_format=torch.channels_last) for _ in range(NC)]
    bias = torch.randn(cout, device=dev).to(dt)
    tok = x.permute(0, 2, 3, 1).reshape(B * h * h, cin).contiguous()
    wp = [pack_conv3x3(w.contiguous()) for w in wt]
    ours = timeit([(lambda w=w: ops.conv3x3(tok, w, B, h, h, cin, bias=bias)) for w in wp])
    lib = timeit([(lambda w=w: F.conv2d(x, w, bias, padding=1)) for w in wt])
    fl = 2.0 * B * h * h * cout * 9 * cin
    print(f"{h}x{h} {cin}->{cout}   {ours:8.1f} ({fl / ours / 1e6:5.0f} TF)   {lib:8.1f} ({fl / lib / 1e6:5.0f} TF)", flush=True)

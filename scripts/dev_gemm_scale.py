"""dev helper: conv 64x64 320->320 throughput vs batch (number of blocks) - is the kernel bound by a shared resource?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
from theatergen_amd.weights_pack import pack_conv3x3
dev, dt = "cuda:0", torch.bfloat16
g = torch.Generator().manual_seed(0)
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (h, cin, cout) in ((64, 320, 320), (32, 640, 640), (16, 1280, 1280)):
    w = pack_conv3x3((torch.randn(cout, cin, 3, 3, generator=g) / 54).to(dt)).to(dev)
    for B in (1, 2, 4, 8, 16, 32):
        x = torch.randn(B * h * h, cin, generator=g).to(dev, dt)
        ms = timeit(lambda: ops.conv3x3(x, w, B, h, h, cin, force_tile=1))
        M = B * h * h
        blocks = ((M + 127) // 128) * ((cout + 127) // 128)
        print(f"conv h={h} {cin}->{cout} B={B:2d} M={M:6d} blocks={blocks:5d}  {2.0*M*cout*9*cin/ms/1e9:6.0f} TF  {ms*1e3:8.1f} us  TF/active-CU={2.0*M*cout*9*cin/ms/1e9/min(blocks,512)*2:5.2f}", flush=True)
M, N = 8192, 4096
for K in (512, 1024, 4096):
    a = torch.randn(M, K, generator=g).to(dev, dt); w = (torch.randn(N, K, generator=g) / 64).to(dev, dt)
    for rows in (512, 1024, 2048, 4096, 8192):
        ms = timeit(lambda: ops.gemm(a, w, rows, N, K, force_tile=1))
        print(f"plain M={rows} N={N} K={K} blocks={rows//128*32}  {2.0*rows*N*K/ms/1e9:6.0f} TF", flush=True)

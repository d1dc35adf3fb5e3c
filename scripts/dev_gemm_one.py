"""dev helper: run ONE GEMM/conv shape a few times (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
from theatergen_amd.weights_pack import pack_conv3x3
dev, dt = "cuda:0", torch.bfloat16
kind = sys.argv[1]; ft = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = torch.Generator().manual_seed(0)
if kind == "conv":
    B, h, cin, cout = 16, 64, 320, 320
    x = torch.randn(B * h * h, cin, generator=g).to(dev, dt)
    w = pack_conv3x3((torch.randn(cout, cin, 3, 3, generator=g) / 54).to(dt)).to(dev)
    fn = lambda: ops.conv3x3(x, w, B, h, h, cin, force_tile=ft)
else:
    M, N, K = 8192, 4096, 4096
    a = torch.randn(M, K, generator=g).to(dev, dt); w = (torch.randn(N, K, generator=g) / 64).to(dev, dt)
    fn = lambda: ops.linear(a, w, force_tile=ft)
for _ in range(5):
    fn()
torch.cuda.synchronize()

"""dev helper: ONE forced (tile, split) configuration of the 12 x 12-level conv (M = 288), for scripts/dev_conv12_all.sh (each configuration in its own
process under `timeout`, so a configuration that does not finish cannot take the sweep with it)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
ft, fs, c1 = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev, dt = "cuda:0", torch.bfloat16
B, h, cin, cout = 2, 12, 1280, 1280
g = torch.Generator().manual_seed(0)
ctot = cin + c1; M = B * h * h
X0 = torch.randn(M, cin, generator=g).to(dev, dt); X1 = torch.randn(M, c1, generator=g).to(dev, dt) if c1 else None
Ws = [(torch.randn(cout, 9 * ctot, generator=g) / math.sqrt(9 * ctot)).to(dev, dt) for _ in range(4)]
bias = torch.randn(cout, generator=g).to(dev, dt)
pl = ops.conv3x3(X0, Ws[0], B, h, h, cin, x1=X1, c1=c1, bias=bias, force_tile=ft, force_split_k=fs, plan_only=True)
print(f"ft={ft} fs={fs} c1={c1} plan={pl}", flush=True)
ref = ops.conv3x3(X0, Ws[0], B, h, h, cin, x1=X1, c1=c1, bias=bias).float()
torch.cuda.synchronize()
out = ops.conv3x3(X0, Ws[0], B, h, h, cin, x1=X1, c1=c1, bias=bias, force_tile=ft, force_split_k=fs).float()
torch.cuda.synchronize()
err = ((out - ref).norm() / ref.norm()).item()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(16):
    ops.conv3x3(X0, Ws[i % 4], B, h, h, cin, x1=X1, c1=c1, bias=bias, force_tile=ft, force_split_k=fs)
e1.record(); torch.cuda.synchronize()
print(f"ft={ft} fs={fs} c1={c1} {e0.elapsed_time(e1) / 16 * 1e3:.1f} us rel-err-vs-default {err:.2e}", flush=True)

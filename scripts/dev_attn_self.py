"""dev helper (not part of the product or the tests): time the level-0 self-attention of the SD-1.5 CFG-batch-16 call
(4096 queries x 4096 keys, 8 heads x 40, 128 (batch, head) pairs); PMC target for scripts/dev_attn_pmc.sh.
Usage: python scripts/dev_attn_self.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from theatergen_amd import ops

dev = "cuda:0"
B, H, D, L = 16, 8, 40, 4096
C = H * D
g = torch.Generator().manual_seed(0)
q = torch.randn(B * L, C, generator=g).to(dev, torch.bfloat16)
k = torch.randn(B * L, C, generator=g).to(dev, torch.bfloat16)
vt = torch.randn(B, C, L, generator=g).to(dev, torch.bfloat16)
out = torch.empty(B * L, C, device=dev, dtype=torch.bfloat16)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def run():
    ops.attention(q, C, L * C, k, C, L * C, vt, L, C * L, L, B, H, D, L, D ** -0.5, out, C, L * C)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
flops = 4.0 * L * L * D * B * H
print(f"self-attention {B}x{H}x{L}x{L} d={D}: {us:8.1f} us  {flops / us / 1e6:6.0f} TF useful")

# level-0 IP-Adapter cross-attention: 4096 queries x (77 text + 4 image) keys
Lk, Li = 77, 4
kc = torch.randn(B * Lk, C, generator=g).to(dev, torch.bfloat16)
vtc = torch.randn(B, C, 80, generator=g).to(dev, torch.bfloat16)          # V^T rows padded to a 16-byte pitch
ki = torch.randn(B * Li, C, generator=g).to(dev, torch.bfloat16)
vti = torch.randn(B, C, 8, generator=g).to(dev, torch.bfloat16)


def run_cross():
    ops.attention(q, C, L * C, kc, C, Lk * C, vtc, 80, C * 80, Lk, B, H, D, L, D ** -0.5, out, C, L * C,
                  k1=ki, k1_ld=C, k1_bs=Li * C, vt1=vti, vt1_ld=8, vt1_bs=C * 8, len1=Li, w1=1.0)


for _ in range(3):
    run_cross()
torch.cuda.synchronize()
e0.record()
for _ in range(iters):
    run_cross()
e1.record()
torch.cuda.synchronize()
print(f"cross-attention 4096 x (77 + 4): {e0.elapsed_time(e1) / iters * 1e3:8.1f} us")

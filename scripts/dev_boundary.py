"""dev helper (round 5): the UNet's two boundary convolutions at the SD-1.5 bench shape (CFG batch 16, 64 x 64): matrix-core kernels vs the fp32-FMA / dot-product
kernels (TG_CONV_IN_MFMA / TG_CONV_OUT_MFMA = 0 in a second process), HIP-event us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
from theatergen_amd.weights_pack import pack_conv3x3
dev, dt = "cuda:0", torch.bfloat16
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B, h, w = 16, 64, 64
smp = torch.randn(B, 4, h, w, device=dev)
w_in = pack_conv3x3((torch.randn(320, 4, 3, 3) / 6).to(dt)).to(dev)
b_in = torch.randn(320, device=dev).to(dt)
x = torch.randn(B * h * w, 320, device=dev).to(dt)
gamma, beta = torch.ones(320, device=dev).to(dt), torch.zeros(320, device=dev).to(dt)
w_out = pack_conv3x3((torch.randn(4, 320, 3, 3) / 54).to(dt)).to(dev)
b_out = torch.randn(4, device=dev).to(dt)
print("knobs", {k: os.environ.get(k) for k in ("TG_CONV_IN_MFMA", "TG_CONV_OUT_MFMA")})
print("conv_in            %7.1f us" % timeit(lambda: ops.conv_in(smp, w_in, b_in, 320, dt)))
y = ops.groupnorm(x, B, h * w, 32, 1e-5, gamma, beta, silu=True)
print("conv_out (plain)   %7.1f us" % timeit(lambda: ops.conv_out(y, w_out, b_out, B, h, w, 4, torch.float32)))
print("groupnorm apply    %7.1f us" % timeit(lambda: ops.groupnorm(x, B, h * w, 32, 1e-5, gamma, beta, silu=True)))
print("groupnorm coef     %7.1f us" % timeit(lambda: ops.groupnorm_coef(x, B, h * w, 32, 1e-5, gamma, beta)))
if ops.conv_out_takes_gn(320, h, w, 4):
    coef = ops.groupnorm_coef(x, B, h * w, 32, 1e-5, gamma, beta)
    print("conv_out (GN fused)%7.1f us" % timeit(lambda: ops.conv_out(x, w_out, b_out, B, h, w, 4, torch.float32, coef=coef, silu=True)))

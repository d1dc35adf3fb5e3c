"""dev helper (not part of the product or the tests): time tg_gemm on UNet-representative shapes, per tile config
(force_tile: 1 + tile id; 0 = the heuristic incl. the LDS-halo conv kernel and the tail split), with a quick correctness check
against torch on the GPU.  Usage: python scripts/dev_gemm_bench.py [--quick]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from theatergen_amd import ops
from theatergen_amd.weights_pack import pack_conv3x3

dev = "cuda:0"
dt = torch.bfloat16
B = 16


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def bench_plain(M, N, K, variants, with_res=False):
    g = torch.Generator().manual_seed(0)
    a = (torch.randn(M, K, generator=g)).to(dev, dt)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev, dt)
    bias = torch.randn(N, generator=g).to(dev, dt)
    res = torch.randn(M, N, generator=g).to(dev, dt) if with_res else None
    ref = (a.float() @ w.float().t() + bias.float())
    if with_res:
        ref = ref + res.float()
    row = [f"plain{'+res' if with_res else '    '} M={M:6d} N={N:5d} K={K:5d}"]
    for name, ft in variants:
        try:
            out = ops.linear(a, w, bias, res=res, force_tile=ft)
            err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
            ms = timeit(lambda: ops.linear(a, w, bias, res=res, force_tile=ft))
            row.append(f"{name}:{2.0 * M * N * K / ms / 1e9:6.0f}TF{'' if err < 2e-2 else ' ERR%.3f' % err}")
        except RuntimeError as e:
            row.append(f"{name}: fail {str(e)[:40]}")
    print("  ".join(row), flush=True)


def bench_conv(h, cin, cout, variants, c1=0, stride=1, up=False):
    g = torch.Generator().manual_seed(0)
    ctot = cin + c1
    x = torch.randn(B, ctot, h, h, generator=g).to(dev, dt)
    wt = (torch.randn(cout, ctot, 3, 3, generator=g) / (9 * ctot) ** 0.5).to(dev, dt)
    bias = torch.randn(cout, generator=g).to(dev, dt)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up else x.float()
    ref = F.conv2d(xin, wt.float(), bias.float(), stride=stride, padding=1)
    oh = ref.shape[-1]
    ref = ref.permute(0, 2, 3, 1).reshape(-1, cout)
    tok = x.permute(0, 2, 3, 1).reshape(B * h * h, ctot)
    x0 = tok[:, :cin].contiguous()
    x1 = tok[:, cin:].contiguous() if c1 else None
    wp = pack_conv3x3(wt)
    M, K = B * oh * oh, 9 * ctot
    row = [f"conv  h={h:3d} {cin:4d}+{c1:4d}->{cout:4d} s{stride}{'u' if up else ' '} M={M:6d} K={K:5d}"]
    for name, ft in variants:
        try:
            fn = lambda: ops.conv3x3(x0, wp, B, h, h, cin, x1=x1, c1=c1, stride=stride, upsample=up, bias=bias, force_tile=ft)
            out = fn()
            err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
            ms = timeit(fn)
            row.append(f"{name}:{2.0 * M * cout * K / ms / 1e9:6.0f}TF{'' if err < 2e-2 else ' ERR%.3f' % err}")
        except RuntimeError as e:
            row.append(f"{name}: fail {str(e)[:40]}")
    print("  ".join(row), flush=True)


if __name__ == "__main__":
    variants = [("auto(halo)", 0), ("128s2", 1)]
    if "--quick" in sys.argv:
        variants = [("auto", 0), ("128x128", 1), ("64x64", 2), ("128x128 3-stage", 5)]
    bench_conv(64, 320, 320, variants)
    bench_conv(64, 640, 320, variants, c1=320)
    bench_conv(32, 640, 640, variants)
    bench_conv(32, 1280, 640, variants, c1=640)
    bench_conv(16, 1280, 1280, variants)
    bench_conv(16, 1280, 1280, variants, c1=1280)
    bench_conv(8, 1280, 1280, variants)
    bench_conv(32, 640, 640, variants, up=True)
    bench_conv(64, 320, 320, variants, stride=2)
    bench_plain(65536, 960, 320, variants)
    bench_plain(65536, 320, 320, variants)
    bench_plain(65536, 320, 320, variants, with_res=True)
    bench_plain(65536, 320, 1280, variants, with_res=True)
    bench_plain(65536, 2560, 320, variants)
    bench_plain(65536, 320, 1280, variants)
    bench_plain(16384, 1920, 640, variants)
    bench_plain(16384, 5120, 640, variants)
    bench_plain(16384, 640, 2560, variants)
    bench_plain(4096, 3840, 1280, variants)
    bench_plain(4096, 10240, 1280, variants)
    bench_plain(4096, 1280, 5120, variants)
    bench_plain(1024, 1280, 1280, variants)
    bench_plain(16, 20160, 1280, variants)
    bench_plain(8192, 4096, 4096, variants)

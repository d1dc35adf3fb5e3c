"""dev timing (not part of the product): the ping-pong 256 x 256 GEMM (tg_gemm force_tile 24) against what the planner picked before round 6 (TG_PP=0) and the
vendor's bare matmul, isolated launches, rotating weights, random operands."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
from theatergen_amd.weights_pack import pack_geglu, pack_ln_linear

dev, dt = "cuda:0", torch.bfloat16


def timeit(fns, iters=40):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for i in range(iters):
            fns[i % len(fns)]()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def old(fn):
    os.environ["TG_PP"] = "0"
    try:
        return fn()
    finally:
        os.environ.pop("TG_PP", None)


print("shape / epilogue                          pp (us, TF)      before round 6 (us, TF)   vendor bare matmul (us, TF)")
for (M, N, K, kind) in [(16384, 640, 640, "res"), (16384, 640, 2560, "res"), (16384, 640, 1920, "plain"), (16384, 1920, 640, "qkv_ln"), (65536, 320, 1280, "res"), (65536, 320, 640, "plain"),
                        (16384, 5120, 640, "geglu"), (4096, 10240, 1280, "geglu"), (65536, 2560, 320, "geglu"), (4096, 3840, 1280, "qkv_ln"), (16384, 1920, 640, "plain"),
                        (4096, 1280, 1280, "res"), (16384, 5120, 640, "plain"), (8192, 4096, 4096, "plain"), (4096, 1280, 5120, "res"), (16384, 1280, 2560, "res"),
                        (2048, 10240, 1280, "geglu"), (8192, 5120, 640, "geglu"), (2048, 1280, 5120, "res"), (8192, 640, 2560, "res")]:
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.rand(M, K, generator=g) * 2 - 1).to(dt).to(dev)
    ws = [((torch.rand(N, K, generator=g) * 2 - 1) / math.sqrt(K)).to(dt).to(dev) for _ in range(4)]
    bias = torch.randn(N, generator=g).to(dt).to(dev)
    fl = 2.0 * M * N * K
    if kind == "geglu":
        packed = [pack_geglu(w, bias) for w in ws]
        mk = lambda ft: [(lambda wp=wp, bp=bp: ops.gemm(a, wp, M, N, K, bias=bp, geglu=True, force_tile=ft)) for wp, bp in packed]
    elif kind == "res":
        res = torch.randn(M, N, generator=g).to(dt).to(dev)
        mk = lambda ft: [(lambda w=w: ops.linear(a, w, bias, res=res, force_tile=ft)) for w in ws]
    elif kind == "qkv_ln":
        C = K
        gamma, beta = torch.ones(C, device=dev, dtype=dt), torch.zeros(C, device=dev, dtype=dt)
        packed = [pack_ln_linear(w, None, gamma, beta) for w in ws]
        B = 16
        rows = M // B
        out = torch.empty((M, 2 * C), dtype=dt, device=dev)
        out_t = torch.empty((B, C, rows), dtype=dt, device=dev)
        st = ops.layernorm_stats(a, 1e-5)

        def mk(ft):
            if ft in (24, 25):
                return [(lambda p=p: ops.gemm(a, p[0], M, N, K, rows_per_batch=rows, out=out, n_split=2 * C, out_t=out_t, ldt=rows, ln=(p[1], p[2], 1e-5, ops.layernorm_stats(a, 1e-5)), force_tile=ft)) for p in packed]
            return [(lambda p=p: ops.gemm(a, p[0], M, N, K, rows_per_batch=rows, out=out, n_split=2 * C, out_t=out_t, ldt=rows, ln=(p[1], p[2], 1e-5))) for p in packed]
    else:
        mk = lambda ft: [(lambda w=w: ops.linear(a, w, force_tile=ft)) for w in ws]
    ft_pp = 24 if N % 256 == 0 else 25
    try:
        t_pp = timeit(mk(ft_pp))
    except RuntimeError as e:
        t_pp = float("nan")
    t_old = old(lambda: timeit(mk(0)))
    o2 = torch.empty(M, N, device=dev, dtype=dt)
    t_v = timeit([(lambda w=w: torch.matmul(a, w.t(), out=o2)) for w in ws])
    print(f"M={M:6d} N={N:5d} K={K:5d} {kind:7s}   {t_pp:7.1f} {fl / t_pp / 1e6:5.0f}      {t_old:7.1f} {fl / t_old / 1e6:5.0f}      {t_v:7.1f} {fl / t_v / 1e6:5.0f}", flush=True)

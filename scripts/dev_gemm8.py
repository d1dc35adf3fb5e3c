"""dev probe driver (not part of the product): the 8-wave ping-pong GEMM loop (scripts/dev_gemm8.hip) against torch.matmul (vendor) and tg_gemm on one box."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "_build", "libgemm8.so"))
lib.gemm8.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
dev, dt = "cuda:0", torch.bfloat16
st = torch.cuda.current_stream().cuda_stream


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


def g8(bn, flags, a, w, o):
    M, K = a.shape
    N = w.shape[0]
    rc = lib.gemm8(bn, flags, a.data_ptr(), w.data_ptr(), o.data_ptr(), M, N, K, st)
    assert rc == 0, rc


variants = [v.split(":") for v in os.environ.get("G8_VARIANTS", "256:0,256:1,128:0,128:1").split(",")]
shapes = [(512, 512, 256), (8192, 4096, 4096), (4096, 4096, 4096), (16384, 5120, 640), (4096, 10240, 1280), (4096, 1280, 5120), (16384, 640, 2560), (16384, 1920, 640), (4096, 3840, 1280)]
for (M, N, K) in shapes:
    a = (torch.rand(M, K, device=dev) * 2 - 1).to(dt)
    ws = [((torch.rand(N, K, device=dev) * 2 - 1) / K ** 0.5).to(dt) for _ in range(4)]
    ref = (a.float() @ ws[0].float().t())
    line = f"M={M:6d} N={N:5d} K={K:5d}  "
    fl = 2.0 * M * N * K
    for bn, flags in variants:
        bn, flags = int(bn), int(flags)
        if N % bn or M % 256:
            line += f" g8<{bn},{flags}>   n/a          "
            continue
        o = torch.zeros(M, N, device=dev, dtype=dt)
        g8(bn, flags, a, ws[0], o)
        torch.cuda.synchronize()
        err = ((o.float() - ref).norm() / ref.norm()).item()
        mx = (o.float() - ref).abs().max().item()
        if M <= 512:
            line += f" g8<{bn},{flags}> rel {err:.2e} max {mx:.2e} |"
            continue
        t = timeit([(lambda w=w: g8(bn, flags, a, w, o)) for w in ws])
        line += f" g8<{bn},{flags}> {t:7.1f} us {fl / t / 1e6:5.0f} TF (rel {err:.1e}) |"
    if M > 512:
        o2 = torch.empty(M, N, device=dev, dtype=dt)
        t = timeit([(lambda w=w: torch.matmul(a, w.t(), out=o2)) for w in ws])
        line += f" vendor {t:7.1f} us {fl / t / 1e6:5.0f} TF |"
        t = timeit([(lambda w=w: ops.linear(a, w)) for w in ws])
        line += f" tg_gemm {t:7.1f} us {fl / t / 1e6:5.0f} TF"
    print(line, flush=True)

"""dev helper: print the plan (tile, tail split, kernel) tg_gemm picks for the SD-1.5 UNet's GEMM / conv shapes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import _lib
L = _lib.lib()
def plan(mode, M, N, K, conv=None, c0=None):
    d = _lib.GemmDesc()
    d.dtype = 0; d.mode = mode; d.M, d.N, d.K = M, N, K
    d.c0 = c0 if c0 else (K if mode == 0 else K // 9)
    x = torch.empty(16, device="cuda", dtype=torch.bfloat16)
    d.a0 = d.w = d.out = x.data_ptr(); d.ldc = N; d.out_scale = 1.0
    if conv: d.batch, d.in_h, d.in_w, d.out_h, d.out_w, d.stride, d.upsample = conv
    tm, tn, sp, kk = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    _lib.check(L.tg_gemm_plan(C.byref(d), C.byref(tm), C.byref(tn), C.byref(sp), C.byref(kk)))
    tiles = -(-M // tm.value) * -(-N // tn.value)
    return f"tile {tm.value}x{tn.value} tiles={tiles:5d} split={sp.value} kind={kk.value} ws={L.tg_gemm_workspace_bytes(C.byref(d)) / 1e6:.1f}MB"
B = 16
for (h, c) in [(64, 320), (32, 640), (16, 1280), (8, 1280)]:
    M = B * h * h
    print(f"--- {h}x{h} C={c} M={M}")
    for name, N, K in [("proj", c, c), ("qkv", 3 * c, c), ("ff1", 8 * c, c), ("ff2", c, 4 * c)]:
        print(f"  gemm {name:5s} N={N:5d} K={K:5d}: {plan(0, M, N, K)}")
    for cin in (c, 2 * c):
        print(f"  conv {cin:4d}->{c:4d}: {plan(1, M, c, 9 * cin, conv=(B, h, h, h, h, 1, 0), c0=cin)}")

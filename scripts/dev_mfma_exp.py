"""dev probe driver (not part of the product): scripts/dev_mfma_exp.hip — do MFMA and v_exp_f32 / v_fma_f32 overlap on a gfx950 SIMD?  ns per iteration and wave."""
import ctypes
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "_build", "libmfma_exp.so"))
lib.mfma_exp_probe.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = "cuda:0"
src = torch.rand(4096, device=dev) * 2 - 1
out = torch.zeros(4, device=dev)
st = torch.cuda.current_stream().cuda_stream
ROUNDS = 200000


def t_ns(nm, ne, layout, kind, threads):
    def go():
        rc = lib.mfma_exp_probe(nm, ne, layout, kind, threads, src.data_ptr(), ROUNDS, out.data_ptr(), st)
        assert rc == 0, (rc, nm, ne, layout, kind, threads)
    go()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        go()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e6 / ROUNDS)
    return best


res = {}
print("ns per iteration (8 MFMA 32x32x16 = 256 matrix-pipe cycles; v_exp_f32 quarter rate = 16 cycles each at one wave)", flush=True)
for threads in (256, 512, 768):
    for ne in (8, 16, 32):
        m = t_ns(8, 0, 0, 0, threads)
        e = t_ns(0, ne, 0, 0, threads)
        both0 = t_ns(8, ne, 0, 0, threads)
        both1 = t_ns(8, ne, 1, 0, threads)
        line = dict(waves_per_simd=threads // 256, n_exp=ne, mfma_only=m, exp_only=e, both_blocked=both0, both_interleaved=both1)
        if threads == 512:
            line["role_split"] = t_ns(8, ne, 2, 0, threads)
        res[f"{threads}/{ne}"] = line
        print(json.dumps(line), flush=True)
for ne in (16, 32):
    m = t_ns(8, 0, 0, 0, 256)
    f = t_ns(0, ne, 0, 1, 256)
    line = dict(waves_per_simd=1, n_fma=ne, mfma_only=m, fma_only=f, both_blocked=t_ns(8, ne, 0, 1, 256), both_interleaved=t_ns(8, ne, 1, 1, 256),
                role_split_2waves=t_ns(8, ne, 2, 1, 512))
    res[f"fma/{ne}"] = line
    print(json.dumps(line), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/r6_mfma_exp_overlap.json", "w"), indent=1)

"""dev probe driver (not part of the product): sustained matrix-pipe throughput on RANDOM bf16, every CU busy, >= 2.5 s per arm,
rocm-smi sampled from a side thread -> gpurun_out/r6_mfma_sustained.json (copied to profiles/).  The vendor GEMM (torch.matmul -> hipBLASLt) runs
beside it on the same box, same data fills, as a yardstick only."""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import ClockPowerSampler

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "_build", "libmfma_sustained.so"))
lib.probe_sustain.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
lib.probe_sustain_flop.restype = ctypes.c_long
lib.probe_sustain_flop.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
dev = "cuda:0"
SECONDS = float(os.environ.get("SUSTAIN_SECONDS", "2.5"))


def run_for(fn, flop_per_call, seconds=SECONDS):
    """calls fn back to back for `seconds`; TFLOP/s over the whole window and over its second half, with the clock / power medians under load"""
    fn()
    torch.cuda.synchronize()
    stamps = []
    with ClockPowerSampler(period=0.25) as cp:
        t0 = time.time()
        e_prev = torch.cuda.Event(enable_timing=True)
        e_prev.record()
        n = 0
        while time.time() - t0 < seconds:
            for _ in range(8):
                fn()
            n += 8
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            e.synchronize()
            stamps.append((n, e_prev.elapsed_time(e)))
        torch.cuda.synchronize()
    total_ms = stamps[-1][1]
    half = [s for s in stamps if s[1] >= total_ms / 2]
    n_half = stamps[-1][0] - half[0][0]
    ms_half = total_ms - half[0][1]
    r = {"tflops": round(flop_per_call * stamps[-1][0] / total_ms / 1e9, 1), "tflops_second_half": round(flop_per_call * n_half / max(ms_half, 1e-6) / 1e9, 1),
         "seconds": round(total_ms / 1e3, 2), "calls": stamps[-1][0]}
    r.update(cp.summary())
    return r


out = {"device": torch.cuda.get_device_name(0), "seconds_per_arm": SECONDS, "arms": {}}
src_rand = (torch.rand(64 << 20, device=dev) * 2 - 1).to(torch.bfloat16)          # uniform [-1, 1): 128 MB
src_zero = torch.zeros_like(src_rand)
sink = torch.zeros(16, device=dev)
st = torch.cuda.current_stream().cuda_stream
blocks = torch.cuda.get_device_properties(0).multi_processor_count
for wps in (2, 1):
    for mode in (0, 1, 2):
        for name, src in (("random", src_rand), ("zeros", src_zero)):
            rounds = 4096
            flop = lib.probe_sustain_flop(wps, rounds, blocks)

            def fn(mode=mode, wps=wps, src=src, rounds=rounds):
                rc = lib.probe_sustain(mode, wps, src.data_ptr(), src.numel() // 8, rounds, blocks, sink.data_ptr(), st)
                assert rc == 0, rc
            key = f"mfma_loop mode{mode} ({['mfma only', '+ LDS fragment reads', '+ LDS reads + LDS-DMA'][mode]}) {wps} wave/SIMD {name}"
            out["arms"][key] = run_for(fn, flop)
            print(key, out["arms"][key], flush=True)

# vendor GEMM beside it (yardstick only)
for (M, N, K) in [(8192, 4096, 4096), (16384, 640, 2560), (4096, 1280, 5120), (4096, 10240, 1280), (16384, 5120, 640)]:
    for name in ("random", "zeros"):
        if name == "zeros" and M != 8192:
            continue
        a = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16) if name == "random" else torch.zeros(M, K, device=dev, dtype=torch.bfloat16)
        ws = [((torch.rand(N, K, device=dev) * 2 - 1) / K ** 0.5).to(torch.bfloat16) if name == "random" else torch.zeros(N, K, device=dev, dtype=torch.bfloat16) for _ in range(4)]
        o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        i = [0]

        def fn(a=a, ws=ws, o=o, i=i):
            torch.matmul(a, ws[i[0] & 3].t(), out=o)
            i[0] += 1
        key = f"vendor matmul {M}x{N}x{K} {name}"
        out["arms"][key] = run_for(fn, 2.0 * M * N * K)
        out["arms"][key]["us_per_call"] = round(out["arms"][key]["seconds"] * 1e6 / out["arms"][key]["calls"], 1)
        print(key, out["arms"][key], flush=True)

os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/r6_mfma_sustained.json", "w") as f:
    json.dump(out, f, indent=1)
print("wrote gpurun_out/r6_mfma_sustained.json")

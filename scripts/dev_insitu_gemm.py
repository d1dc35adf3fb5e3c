"""dev helper: IN-SITU per-launch times of the GEMM / conv launches of one eager CFG-batch-16 UNet step (HIP events around every
tg_gemm launch, operands as the real call leaves them in L2 / MALL), for several TG_GEMM_FLAGS variants in ONE process,
interleaved; prints per (kernel, shape, epilogue) the mean microseconds of each variant.
    python scripts/dev_insitu_gemm.py "8 0" [reps]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from theatergen_amd import ops
from theatergen_amd.pipelines import DenoiseEngine

variants = (sys.argv[1] if len(sys.argv) > 1 else "8 0").split()
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
cfg, sd, unet, adapter = bench.build_model("sd15", torch.bfloat16, dev)
eng = DenoiseEngine(unet, None, n_img=8, height=512, width=512, num_inference_steps=50, guidance_scale=7.5, enc_len=81, use_graph=False)
g = torch.Generator().manual_seed(0)
eng.set_conditioning((torch.randn(16, 81, 768, generator=g) * 0.5).to(dev, torch.bfloat16))
lat = torch.randn(8, 4, 64, 64, generator=g)
acc = {v: collections.OrderedDict() for v in variants}
tot = {v: 0.0 for v in variants}
with torch.no_grad():
    eng._reset(lat)
    for v in variants:
        os.environ["TG_GEMM_FLAGS"] = v
        eng._step()
    torch.cuda.synchronize()
    for r in range(reps):
        for v in variants:
            os.environ["TG_GEMM_FLAGS"] = v
            ops.gemm_profile_start()
            eng._step()
            torch.cuda.synchronize()
            for i, rec in enumerate(ops.gemm_profile_stop()):
                key = (i, rec["M"], rec["N"], rec["K"])
                a = acc[v].setdefault(key, [rec["kernel"], 0.0])
                a[1] += rec["ms"] * 1e3 / reps
                tot[v] += rec["ms"] / reps
# aggregate by (shape) over launch index
agg = collections.OrderedDict()
for v in variants:
    for (i, M, N, K), (kern, us) in acc[v].items():
        a = agg.setdefault((M, N, K, acc[variants[0]][(i, M, N, K)][0]), {vv: [0.0, 0, ""] for vv in variants})
        a[v][0] += us
        a[v][1] += 1
        a[v][2] = kern
print("variants:", variants, " total GEMM-family ms per call:", {v: round(t, 3) for v, t in tot.items()})
for (M, N, K, k0), d in sorted(agg.items(), key=lambda kv: -kv[1][variants[0]][0]):
    cells = "  ".join(f"{v}: {d[v][0] / max(d[v][1], 1):7.1f}us x{d[v][1]:3d} [{d[v][2][:28]}]" for v in variants)
    print(f"M={M:6d} N={N:5d} K={K:5d}  {cells}")

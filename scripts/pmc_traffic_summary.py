"""gpurun_out/pmc_traffic/summary.json (written by scripts/pmc_traffic.sh on the GPU box) -> profiles/r1_pmc_traffic.json,
the per-kernel-family HBM-side traffic bench.py quotes in `roofline.traffic`.
FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM section)."""
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = json.load(open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(R, "gpurun_out/pmc_traffic/summary.json")))
FAMILIES = [
    ("gemm_glds_kernel<plain,128x128>", lambda n: "gemm_glds_kernel" in n and "Li128ELi128ELi2ELi2ELb0E" in n),
    ("gemm_glds_kernel<conv,128x128>", lambda n: "gemm_glds_kernel" in n and "Li128ELi128ELi2ELi2ELb1E" in n),
    ("conv_halo_kernel<128x128>", lambda n: "conv_halo_kernel" in n),
    ("groupnorm", lambda n: "gn_apply_kernel" in n or "gn_partial_kernel" in n),
    ("layernorm", lambda n: "layernorm_kernel" in n),
    ("attention_kernel", lambda n: "attention_kernel" in n),
]
out = {
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (one counter per pass) -- python bench.py --steps 1 "
              "--warmup 0 --ddim-steps 3 --no-cpu-baseline --no-roofline (scripts/pmc_traffic.sh, scripts/pmc_traffic_summary.py)",
    "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE checked against LayerNorm "
                  "(65536x320 bf16 = 40960 KiB reported)",
    "note": "memory-side L2 requests: Infinity-Cache hits are counted, so this is an upper bound on HBM bytes",
    "kernels": {},
}
for label, match in FAMILIES:
    f = [v for k, v in src["FETCH_SIZE"].items() if match(k)]
    w = [v for k, v in src["WRITE_SIZE"].items() if match(k)]
    if not f or not w:
        continue
    launches = int(sum(v["launches"] for v in f))
    assert launches == int(sum(v["launches"] for v in w)), label
    fr, wr = sum(v["sum"] for v in f), sum(v["sum"] for v in w)
    fb, wb = int(fr * 1024 * 2 / launches), int(wr * 1024 / launches)
    out["kernels"][label] = {"launches": launches, "FETCH_SIZE_KiB_raw": fr, "WRITE_SIZE_KiB_raw": wr,
                             "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "traffic_bytes_per_launch": fb + wb}
json.dump(out, open(os.path.join(R, "profiles/r1_pmc_traffic.json"), "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{k:34s} launches {v['launches']:5d}  traffic/launch {v['traffic_bytes_per_launch'] / 1e6:8.1f} MB")

"""gpurun_out/r6prof/* (scripts/r6_profiles.sh on the GPU box) -> profiles/r6_bench_kernel_stats.csv, profiles/r6_pmc_traffic.json
(per kernel family: launches, HBM-side fetch / write bytes per launch with the gfx950 x2 FETCH_SIZE correction) and
profiles/r6_pmc_sq.json (per kernel family: mean SQ counters per launch, MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES /
(SQ_BUSY_CYCLES-normalised SIMD cycles))."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

O = sys.argv[1]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(R, "gpurun_out", "r6prof_out")
os.makedirs(P, exist_ok=True)
# identity of the library the counters were collected on: theatergen_amd/lib/build_info.json travels with the .so (no .git on the GPU box)
try:
    _bi = json.load(open(os.path.join(R, "theatergen_amd", "lib", "build_info.json")))
except (OSError, ValueError):
    _bi = {}
commit = _bi.get("commit", "n/a")
sources_sha256 = _bi.get("sources_sha256")

def _glds128(n):
    return "gemm_glds_kernel" in n and "Li128ELi128ELi2ELi2ELb0E" in n


FAMILIES = [
    # labels = the kernel labels of bench.py's roofline leg (ops.gemm profiling mode), so that `roofline.traffic` finds its kernel
    # template tail since round 5: ..., EPI, LN, XA> -> "ELi<epi>ELi<ln>ELi<xa>EEEvNS_10GemmParamsE"
    ("gemm_glds_kernel<plain,128x128>", lambda n: _glds128(n) and "ELi0ELi0EEEvNS_10GemmParamsE" in n),
    ("gemm_glds_kernel<plain+ln,128x128>", lambda n: _glds128(n) and ("ELi1ELi0EEEvNS_10GemmParamsE" in n or "ELi2ELi0EEEvNS_10GemmParamsE" in n)),
    ("gemm_glds_kernel<plain,128x160>", lambda n: "gemm_glds_kernel" in n and "Li128ELi160ELi4ELi1ELb0E" in n and "ELi0ELi0EEEvNS_10GemmParamsE" in n),
    ("gemm_glds_kernel<plain+ln+xattn,128x160>", lambda n: "gemm_glds_kernel" in n and "Li128ELi160ELi4ELi1ELb0E" in n and ("ELi1ELi80EEEv" in n or "ELi1ELi160EEEv" in n)),
    ("gemm_glds_kernel<plain+ln,128x160>", lambda n: "gemm_glds_kernel" in n and "Li128ELi160ELi4ELi1ELb0E" in n and ("ELi1ELi0EEEvNS" in n or "ELi2ELi0EEEvNS" in n)),
    ("bt_gemm_kernel<256x256>", lambda n: "bt_gemm_kernel" in n and "Li256ELi256E" in n),
    ("gemm_glds_kernel<conv,128x128>", lambda n: "gemm_glds_kernel" in n and "Li128ELi128ELi2ELi2ELb1E" in n),
    ("conv_halo_kernel<128x128>", lambda n: "conv_halo_kernel" in n),
    # round 6: the two-waves-per-SIMD slab kernel (template <T, WI, PRO>), the ping-pong GEMMs (template <T, EPI, LN>)
    ("conv_slab_pp_kernel<128x320>+gn", lambda n: "conv_slab_pp_kernel" in n and "ELb1EEEv" in n),
    ("conv_slab_pp_kernel<128x320,w64>", lambda n: "conv_slab_pp_kernel" in n and "Li64E" in n),
    ("conv_slab_pp_kernel<128x320,w32>", lambda n: "conv_slab_pp_kernel" in n and "Li32E" in n),
    ("conv_slab_pp_kernel<128x320,w16,split2>", lambda n: "conv_slab_pp_kernel" in n and "Li16E" in n),
    ("pp_gemm_kernel<256x256,geglu>", lambda n: "pp_gemm_kernel" in n and "pp160" not in n and "Li2ELi" in n),
    ("pp_gemm_kernel<256x256>", lambda n: "pp_gemm_kernel" in n and "pp160" not in n and "Li2ELi" not in n),
    ("pp160_gemm_kernel<256x160>", lambda n: "pp160_gemm_kernel" in n),
    ("conv_slab_kernel<128x320>+gn", lambda n: "conv_slab_kernel" in n and "ELb1ELb" in n),
    ("conv_slab_kernel<128x320,w64>", lambda n: "conv_slab_kernel" in n and "Li64E" in n),
    ("conv_slab_kernel<128x320,w32>", lambda n: "conv_slab_kernel" in n and "Li32E" in n),
    ("conv_slab_kernel<128x320,w16,split2>", lambda n: "conv_slab_kernel" in n and "Li16E" in n),
    ("splitk_reduce_kernel", lambda n: "splitk_reduce_kernel" in n),
    ("attention_kernel<d40>", lambda n: "attention_kernel" in n and "Li48ELi64E" in n),
    ("attention_kernel<other>", lambda n: "attention_kernel" in n and "Li48ELi64E" not in n),
    ("rc_xattn_kernel<77+4>", lambda n: "rc_xattn_kernel" in n),
    ("rc_ff_kernel<320>+proj_out", lambda n: "rc_ff_kernel" in n),
    ("rc_front_kernel<320>", lambda n: "rc_front_kernel" in n),
    ("rc_linear_kernel<320>", lambda n: "rc_linear_kernel" in n),
    ("groupnorm", lambda n: "gn_apply_kernel" in n or "gn_partial_kernel" in n or "gn_small_kernel" in n or "gn_coef_kernel" in n),
    ("layernorm", lambda n: "layernorm_kernel" in n),
]

st = glob.glob(os.path.join(O, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(P, "r6_bench_kernel_stats.csv"))
for plan in ("sd21", "sdxl"):
    sp = glob.glob(os.path.join(O, "stats_" + plan, "**", "*kernel_stats.csv"), recursive=True)
    if sp:
        shutil.copy(sp[0], os.path.join(P, f"r6_bench_{plan}_kernel_stats.csv"))


def collect(dirname):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(O, dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = agg[r["Counter_Name"]][r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return agg


tr = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (one counter per pass) -- python bench.py --steps 1 --warmup 0 "
                "--ddim-steps 3 --no-cpu-baseline --no-roofline (scripts/r6_profiles.sh)", "commit": commit, "sources_sha256": sources_sha256,
      "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section); values are KiB in the raw counters",
      "note": "memory-side L2 requests: Infinity-Cache hits are counted, so this is an upper bound on HBM bytes", "kernels": {}}
fs, ws = collect("FETCH_SIZE").get("FETCH_SIZE", {}), collect("WRITE_SIZE").get("WRITE_SIZE", {})
for label, match in FAMILIES:
    f = [v for k, v in fs.items() if match(k)]
    w = [v for k, v in ws.items() if match(k)]
    if not f or not w:
        continue
    nf, nw = sum(v[1] for v in f), sum(v[1] for v in w)
    fb, wb = int(sum(v[0] for v in f) * 1024 * 2 / nf), int(sum(v[0] for v in w) * 1024 / nw)
    tr["kernels"][label] = {"launches": nf, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "traffic_bytes_per_launch": fb + wb}
json.dump(tr, open(os.path.join(P, "r6_pmc_traffic.json"), "w"), indent=1)

sq = {"source": "rocprofv3 --kernel-trace --pmc <4 SQ counters per pass> over the same command (scripts/r6_profiles.sh)", "commit": commit, "sources_sha256": sources_sha256,
      "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles summed over waves or SEs; SQ_VALU_MFMA_BUSY_CYCLES "
               "counts cycles summed over SIMDs (32 per 32x32x16 bf16 MFMA: SQ_INSTS_MFMA x 32 reproduces it); GRBM_GUI_ACTIVE is summed over the 8 XCDs "
               "(8 x kernel duration x clock); mfma_busy_frac = MFMA_BUSY / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) = matrix-pipe duty over the kernel",
      "kernels": {}}
allc = {}
for d in ("sq1", "sq2", "sq3"):
    for c, per in collect(d).items():
        allc[c] = per
for label, match in FAMILIES:
    row = {}
    for c, per in allc.items():
        v = [x for k, x in per.items() if match(k)]
        if v:
            row[c] = sum(x[0] for x in v) / max(sum(x[1] for x in v), 1)
    if row:
        if "SQ_VALU_MFMA_BUSY_CYCLES" in row and "GRBM_GUI_ACTIVE" in row and row["GRBM_GUI_ACTIVE"] > 0:
            row["mfma_busy_frac"] = row["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * row["GRBM_GUI_ACTIVE"])
        sq["kernels"][label] = {k: (round(v, 4) if k == "mfma_busy_frac" else round(v)) for k, v in row.items()}
json.dump(sq, open(os.path.join(P, "r6_pmc_sq.json"), "w"), indent=1)
print(json.dumps(tr["kernels"], indent=1))
print(json.dumps(sq["kernels"], indent=1))

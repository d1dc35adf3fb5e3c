#!/usr/bin/env python
"""One hipGraph-replayed DDIM step of the bench, kernel by kernel, from a rocprofv3 kernel trace (+ optionally the per-launch shape
records of bench.py's eager roofline step, TG_DUMP_RECS, zipped onto the trace IN LAUNCH ORDER so that every GEMM / conv / attention
launch carries its (M, N, K)).

    cd /tmp && export TMPDIR=/tmp
    TG_DUMP_RECS=$O/recs.json rocprofv3 --kernel-trace --output-format csv -d $O/trace -- \
        python bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-other-configs
    python scripts/step_breakdown.py $O/trace $O/recs.json > profiles/rN_step_breakdown.txt

A step = the kernels from one `conv_in_fast_kernel` launch up to the next one; the step reported is the median-duration one among the
graph-replayed steps (the eager warm-up / roofline steps carry torch kernels and gaps and are skipped by that choice).
"""
import csv
import glob
import json
import os
import re
import statistics
import sys

SHAPED = ("gemm_glds_kernel", "pp_gemm_kernel", "pp160_gemm_kernel", "conv_slab_pp_kernel", "conv_slab_kernel", "conv_halo_kernel", "bt_gemm_kernel", "lc_gemm_kernel", "ws_gemm_kernel", "attention_kernel", "rc_front_kernel", "rc_linear_kernel", "rc_xattn_kernel", "rc_ff_kernel")


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from theatergen_amd.build import _pretty  # noqa: E402  (rocprofv3 reports MANGLED names: the build's decoder makes them readable)


def pretty(name):
    n = _pretty(name) if name.startswith("_Z") else name
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n)


def short(name):
    m = re.match(r"([A-Za-z_0-9]+)", pretty(name))
    return m.group(1) if m else name[:40]


def variant(name):
    """template arguments that matter for reading the table"""
    n = pretty(name)
    b = short(name)
    if b == "attention_kernel":
        m = re.search(r"attention_kernel<[^,]+,\s*(\d+),\s*(\d+)", n)
        return f"attention_kernel<{m.group(1)},{m.group(2)}>" if m else b
    if b == "gemm_glds_kernel":
        m = re.search(r"gemm_glds_kernel<[^,]+,(\d+),(\d+),\d+,\d+,(true|false),\d+,\d+,(\d+),(\d+)(?:,(\d+))?>", n.replace(" ", ""))
        if m:
            xa = f",xattn{m.group(6)}" if m.group(6) not in (None, "0") else ""
            return f"gemm_glds_kernel<{m.group(1)}x{m.group(2)},{'conv' if m.group(3) == 'true' else 'plain'},epi{m.group(4)},ln{m.group(5)}{xa}>"
    return b


def main():
    tdir = sys.argv[1]
    recs = json.load(open(sys.argv[2])) if len(sys.argv) > 2 and os.path.exists(sys.argv[2]) else None
    files = glob.glob(os.path.join(tdir, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit("no *kernel_trace.csv under " + tdir)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if short(r[2]) in ("conv_in_fast_kernel", "conv_in_mfma_kernel", "conv_in_kernel")]
    steps = []
    for a, b in zip(starts[:-1], starts[1:]):
        ks = rows[a:b]
        wall = rows[b][0] - ks[0][0]
        busy = sum(e - s for s, e, _ in ks)
        foreign = sum(1 for _, _, n in ks if "at::" in n or "elementwise_kernel" in n or "Memcpy" in n or "copyBuffer" in n)
        steps.append((wall, busy, foreign, ks))
    clean = [s for s in steps if s[2] == 0 and s[1] > 0.97 * s[0]] or steps
    med = statistics.median_low([s[0] for s in clean])
    wall, busy, foreign, ks = next(s for s in clean if s[0] == med)
    print(f"# one hipGraph-replayed DDIM step of the SD-1.5 bench (CFG batch 16 UNet call + step epilogue) from a rocprofv3 kernel trace; "
          f"{len(clean)} gap-free steps found, the median one shown")
    print(f"kernels {len(ks)}, kernel time {busy / 1e3:.1f} us, wall {wall / 1e3:.1f} us, torch / runtime kernels inside the step: {foreign}")
    fam = {}
    for s, e, n in ks:
        v = fam.setdefault(variant(n), [0, 0])
        v[0] += 1
        v[1] += e - s
    for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:46s} {c:4d} {t / 1e3:9.1f} us {100 * t / busy:5.1f}%")
    # dominant kernel TEMPLATE of the replayed step (all instantiations summed): what bench.py's `roofline.kernel` names (TG_DOMINANT_JSON = output path)
    tpl = {}
    for s, e, n in ks:
        v = tpl.setdefault(short(n), [0, 0])
        v[0] += 1
        v[1] += e - s
    order = sorted(tpl.items(), key=lambda kv: -kv[1][1])
    if os.environ.get("TG_DOMINANT_JSON") and len(order) >= 2:
        (t1, (c1, u1)), (t2, (c2, u2)) = order[0], order[1]
        with open(os.environ["TG_DOMINANT_JSON"], "w") as f:
            json.dump({"template": t1, "launches": c1, "us_per_step": round(u1 / 1e3, 1), "runner_up": t2, "runner_up_launches": c2,
                       "runner_up_us_per_step": round(u2 / 1e3, 1), "within_5pct": bool(u2 >= 0.95 * u1), "step_kernels": len(ks),
                       "step_kernel_time_us": round(busy / 1e3, 1),
                       "source": "median graph-replayed step of a rocprofv3 --kernel-trace of `python bench.py` (scripts/step_breakdown.py)"}, f, indent=1)
    if recs is None:
        return
    shaped = [(s, e, n) for s, e, n in ks if short(n) in SHAPED]
    if len(shaped) != len(recs):
        print(f"# shape records ({len(recs)}) and shaped launches in the step ({len(shaped)}) differ: no per-shape table")
        return
    by = {}
    for (s, e, n), r in zip(shaped, recs):
        fam_r = r["kernel"].split("<", 1)[0]
        if fam_r != short(n):
            print(f"# order mismatch: trace {short(n)} vs record {r['kernel']}: no per-shape table")
            return
        key = (r["kernel"], r["M"], r["N"], r["K"], r["splits"])
        v = by.setdefault(key, [0, 0, 0.0, 0.0])
        v[0] += 1
        v[1] += e - s
        v[2] += r["flops"]
        v[3] += r["ms"]
    print("\n# per shape, IN SITU (graph replay): kernel, M, N, K, splits, launches, avg us in the replayed step, TFLOP/s, total us, share of the "
          "step; last column: the same launches' avg us in the eager roofline step (HIP events)")
    for k, (c, t, fl, ems) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[0]:40s} M={k[1]:6d} N={k[2]:5d} K={k[3]:6d} s={k[4]} x{c:3d} {t / c / 1e3:8.1f} us {fl / t / 1e3:7.1f} TF {t / 1e3:8.1f} us "
              f"{100 * t / busy:5.1f}%   eager {ems / c * 1e3:7.1f} us")


if __name__ == "__main__":
    main()

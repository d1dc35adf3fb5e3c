#!/usr/bin/env python
"""Same-box A/B of LIBRARY BUILDS (old .so vs new .so) and / or environment switches, interleaved, graph-replayed bench.

    python scripts/ab.py --rounds 3 --out gpurun_out/ab_attn.json \
        --variant r3:lib=theatergen_amd/lib/libtheatergen_hip_r3.so --variant new --variant pipe:TG_ATTN_PIPE=1

A variant is  name[:key=value[,key=value...]]  where key `lib` selects the shared library (THEATERGEN_HIP_LIB) and every other
key is an environment variable.  Box-to-box spread of the pool is +-4 %, so only interleaved runs on ONE box decide anything
(a flag inside one binary is NOT an A/B of two builds — both arms carry the new binary's registers).
"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", action="append", required=True)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--out", default=None)
    ap.add_argument("--bench-args", default="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs")
    a = ap.parse_args()
    variants = []
    for v in a.variant:
        name, _, rest = v.partition(":")
        env = {}
        for kv in filter(None, rest.split(",")):
            k, _, val = kv.partition("=")
            if k == "lib":
                k, val = "THEATERGEN_HIP_LIB", os.path.join(ROOT, val) if not os.path.isabs(val) else val
                assert os.path.exists(val), val
            env[k] = val
        variants.append((name, env))
    res = {n: [] for n, _ in variants}
    for r in range(a.rounds):
        for name, env in variants:
            e = dict(os.environ)
            e.pop("THEATERGEN_HIP_LIB", None)
            e.update(env)
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + a.bench_args.split(), env=e, capture_output=True, text=True, timeout=900)
            try:
                d = json.loads(p.stdout.strip().splitlines()[-1])
                res[name].append(d["ms_per_step"])
                print(f"round {r} {name}: {d['value']} {d['unit']}  {d['ms_per_step']} ms/step", flush=True)
            except (IndexError, ValueError, KeyError):
                print(f"round {r} {name}: FAILED\n{p.stderr[-800:]}", flush=True)
    summary = {n: {"ms_per_step": v, "median": statistics.median(v) if v else None} for n, v in res.items()}
    base = summary[variants[0][0]]["median"]
    for n, s in summary.items():
        if s["median"] and base:
            s["speedup_vs_" + variants[0][0]] = round(base / s["median"], 4)
    print(json.dumps(summary, indent=1))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump({"variants": {n: e for n, e in variants}, "bench_args": a.bench_args, "summary": summary}, f, indent=1)


if __name__ == "__main__":
    main()

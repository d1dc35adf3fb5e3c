"""gpurun_out/parity_metrics.jsonl (written by tests/parity_metrics.py during `pytest -m gpu`) -> profiles/r3_parity_metrics.json (argv[3] = another name):
single-kernel checks condensed to the worst case per dtype, every multi-kernel comparison listed with its tolerance."""
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(R, "gpurun_out", "parity_metrics.jsonl")
note = sys.argv[2] if len(sys.argv) > 2 else ""
recs = [json.loads(l) for l in open(src) if l.strip()]
last = {}
for r in recs:                       # the log is appended across runs: keep the newest record of every comparison
    last[(r["what"], r.get("dtype", ""))] = r
recs = list(last.values())
out = {"source": "tests/parity_metrics.py log of `pytest -m gpu` on MI355X" + (" (" + note + ")" if note else "")}
for key, tag in (("single_kernel_bf16", ".bfloat16"), ("single_kernel_fp16", ".float16")):
    ks = [r for r in recs if r["what"].startswith("kernel:") and r.get("dtype", "").endswith(tag)]
    if ks:
        out[key] = {"n": len(ks), "worst_rel_l2": max(r["rel_l2"] for r in ks), "worst_max_rel": max(r["max_rel"] for r in ks)}
out["comparisons"] = [{k: r[k] for k in ("what", "rel_l2", "max_rel", "l2_tol", "max_tol") if k in r}
                      for r in recs if not r["what"].startswith("kernel:")]
name = sys.argv[3] if len(sys.argv) > 3 else "r3_parity_metrics.json"
json.dump(out, open(os.path.join(R, "profiles", name), "w"), indent=1)
print(out.get("single_kernel_bf16"), out.get("single_kernel_fp16"), len(out["comparisons"]))

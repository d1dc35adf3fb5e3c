"""dev helper: group a TG_DUMP_RECS file (bench.py: per-launch GEMM / conv records) by (kernel, M, N, K, splits) -> launches, avg us, TFLOP/s."""
import json
import sys

recs = json.load(open(sys.argv[1]))
by = {}
for r in recs:
    k = (r["kernel"], r["M"], r["N"], r["K"], r["splits"])
    v = by.setdefault(k, [0, 0.0, 0.0])
    v[0] += 1
    v[1] += r["ms"]
    v[2] += r["flops"]
tot = sum(v[1] for v in by.values())
print(f"total {tot:.3f} ms over {len(recs)} launches")
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:44s} M={k[1]:6d} N={k[2]:5d} K={k[3]:6d} s={k[4]} x{v[0]:3d} {v[1] / v[0] * 1e3:8.1f} us  {v[2] / v[1] / 1e9:7.1f} TF  {v[1]:7.3f} ms {100 * v[1] / tot:5.1f}%")

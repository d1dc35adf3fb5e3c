#!/bin/bash
# kernel mix of ONE latent_backward_guidance iteration (768^2 plan): kernel stats of the sd21 bench with 3 and with 23 eager iterations; the difference / 20
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3bwd; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for n in 3 23; do
  TG_GUIDE_ITERS=$n timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/n$n -- python $R/bench.py --plan sd21 --ddim-steps 10 --steps 1 --warmup 1 --no-cpu-baseline > $O/n$n.log 2>&1
done
python - <<PY
import csv, glob
def load(d):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), int(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load("$O/n3"), load("$O/n23")
rows = []
for k, (c, t) in b.items():
    c0, t0 = a.get(k, (0, 0))
    if c - c0 > 0:
        rows.append((k, (c - c0) / 20.0, (t - t0) / 20.0 / 1e3))
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
with open("$O/per_iteration.txt", "w") as f:
    f.write(f"# kernels of one latent_backward_guidance iteration (SD-2.1 plan 768^2, cond-only forward + reverse pass), rocprofv3 stats (23 - 3 iterations) / 20; total {tot/1e3:.2f} ms\n")
    for k, c, t in rows[:40]:
        f.write(f"{k[:110]:110s} {c:7.1f} launches {t:9.1f} us {100*t/tot:5.1f}%\n")
print(open("$O/per_iteration.txt").read()[:6000])
PY

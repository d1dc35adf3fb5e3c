#!/bin/bash
# HBM traffic of the bench's kernels from the L2 fabric-side counters, one counter per pass (guide: FETCH_SIZE and
# WRITE_SIZE do not fit one pass; never combined with other trace domains).  Output: gpurun_out/pmc_traffic/*.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/pmc_traffic; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_traffic/$c -- \
    python $R/bench.py --steps 1 --warmup 0 --ddim-steps 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_traffic/$c.log 2>&1
done
python - <<PY
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("$R/gpurun_out/pmc_traffic/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            a = agg[r["Kernel_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    out[c] = {k: {"sum": v[0], "launches": v[1]} for k, v in agg.items()}
json.dump(out, open("$R/gpurun_out/pmc_traffic/summary.json", "w"), indent=1)
for c in out:
    for k, v in sorted(out[c].items(), key=lambda kv: -kv[1]["sum"])[:8]:
        print(c, k[:80], "avg/launch", v["sum"] / v["launches"], "n", v["launches"])
PY

#!/bin/bash
# round 4, GPU call F: in-situ kernel stats of the row-chain build (rocprofv3 --kernel-trace --stats over the graph-replayed bench)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
TG_RC=$v timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$v -- \
  python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-other-configs > $O/bench_trace$v.log 2>&1
find $O/trace$v -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_rc$v.csv \;
rm -rf $O/trace$v
done
python - <<PY
import csv
for v in (1,0):
    rows=list(csv.DictReader(open("$O/kernel_stats_rc%d.csv"%v)))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    print("TG_RC=%d total kernel ms %.2f"%(v,tot/1e6))
    for r in rows[:14]:
        print("  %-90s calls %5s avg_us %8.1f pct %s"%(r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY

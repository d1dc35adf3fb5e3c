#!/bin/bash
# round 4, GPU call Q: kernel stats of the Resampler (SDXL-Plus / SD-1.5-Plus geometry, batch 2) on the skinny latent path
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
RS_ONLY=${1:-sdxl_plus} RS_MODES=${2:-1} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/scripts/dev_resampler.py > $O/run.log 2>&1
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/trace
tail -3 $O/run.log
python - <<'P'
import csv, os
rows = list(csv.DictReader(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/r4q/kernel_stats.csv"))))
for r in rows[:16]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(6), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(8), r["Percentage"].rjust(7))
P

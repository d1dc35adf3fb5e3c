#!/bin/bash
# round 6: kernel trace of the bench command -> one graph-replayed step, kernel by kernel (profiles/r6_step_breakdown*.txt).  $1 = tag, extra env via the caller.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r6}
O=$R/gpurun_out/${TAG}_trace; P=$R/gpurun_out/${TAG}_out; mkdir -p $O $P; cd /tmp && export TMPDIR=/tmp
TG_DUMP_RECS=$O/recs.json timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-other-configs > $O/trace.log 2>&1
TG_DOMINANT_JSON=$P/${TAG}_dominant_template.json python $R/scripts/step_breakdown.py $O/trace $O/recs.json > $P/${TAG}_step_breakdown.txt 2>&1
find $O -name "*.csv" -size +2M -delete
head -60 $P/${TAG}_step_breakdown.txt

#!/bin/bash
# round 4, GPU call B: in-situ per-shape step breakdown at HEAD + the whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
TG_DUMP_RECS=$O/recs.json timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- \
  python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-other-configs > $O/bench_trace.log 2>&1
python $R/scripts/step_breakdown.py $O/trace $O/recs.json > $O/step_breakdown.txt 2>&1
rm -rf $O/trace
head -60 $O/step_breakdown.txt
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/tests.txt
cat $O/tests.txt

#!/bin/bash
# round 4, GPU call T: HEAD of the round (time-projection table on): whole GPU suite + default bench line + step breakdown
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4t; mkdir -p $O
cd $R
timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/tests.txt
head -9 $O/tests.txt | tail -3
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 400 $O/bench_default.json
cd /tmp && export TMPDIR=/tmp
TG_DUMP_RECS=$O/recs.json timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- \
  python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-other-configs > $O/bench_trace.log 2>&1
python $R/scripts/step_breakdown.py $O/trace $O/recs.json > $O/step_breakdown.txt 2>&1
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/trace
head -4 $O/step_breakdown.txt

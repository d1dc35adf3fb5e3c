#!/bin/bash
# round 4, GPU call C: in-situ per-shape step breakdown at HEAD + new tests
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
TG_DUMP_RECS=$O/recs.json timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- \
  python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-other-configs > $O/bench_trace.log 2>&1
python $R/scripts/step_breakdown.py $O/trace $O/recs.json > $O/step_breakdown.txt 2>&1
find $O/trace -name "*kernel_trace.csv" -exec gzip -9 {} \;
head -90 $O/step_breakdown.txt
cd $R
timeout 900 python -m pytest tests/test_round4_gpu.py -x -q -k "large_mean or eviction" 2>&1 | tail -8

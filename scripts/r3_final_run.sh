#!/bin/bash
# round-3 evidence of the final build: GPU tests, the default bench line, rocprofv3 kernel stats / PMC passes (scripts/r3_profiles.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3final; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 600 $O/bench_default.json
bash scripts/r3_profiles.sh; tail -5 $R/gpurun_out/r3prof/summary.log

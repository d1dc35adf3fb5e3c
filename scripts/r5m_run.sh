#!/bin/bash
# round 5, GPU call M: the round's evidence on the current sources — whole GPU suite, profile passes (scripts/r5_profiles.sh), default bench line, clock / power
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5m; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
bash scripts/r5_profiles.sh > $O/profiles.log 2>&1; tail -45 $O/profiles.log
cd $R; timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; head -c 2500 $O/bench_line.json; echo

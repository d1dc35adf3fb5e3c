#!/bin/bash
# round 4, GPU call L (final library of the round): the suites the last edits touch, the default bench line, PMC passes on this build
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4l; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_rowchain_gpu.py tests/test_backward_gpu.py tests/test_parity_fullsize_gpu.py tests/test_round4_gpu.py -x -q 2>&1 | tail -6 > $O/tests.txt
cat $O/tests.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 400 $O/bench_default.json
bash $R/scripts/r4_profiles.sh > $O/profiles.log 2>&1
tail -3 $O/profiles.log
timeout 900 python bench.py --no-other-configs > $O/bench_after_pmc.json 2> $O/bench_after_pmc.err

#!/bin/bash
# round 4, GPU call M (FINAL sources of the round): row-chain tests, PMC passes, then the default bench line (its roofline.traffic then names this build)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4m; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_rowchain_gpu.py -x -q 2>&1 | tail -4 > $O/tests.txt
cat $O/tests.txt
bash $R/scripts/r4_profiles.sh > $O/profiles.log 2>&1
tail -3 $O/profiles.log
cp $R/gpurun_out/r4prof_out/r4_pmc_traffic.json $R/gpurun_out/r4prof_out/r4_pmc_sq.json $R/profiles/
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.json

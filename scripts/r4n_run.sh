#!/bin/bash
# round 4, GPU call N: do the round-2 / 3 planner switches that lost then pay on the row-chain build?  same-box, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4n; mkdir -p $O
cd $R
run() { env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt; }
for i in 1 2; do
  run TG_NOOP=1
  run TG_T7_FIT=1
  run TG_T7_MAXK=1280
  run TG_GEMM_FLAGS=256
  run TG_T3_MAX=640
done

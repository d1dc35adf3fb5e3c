#!/bin/bash
# round 4, GPU call I: same-box interleaved A/B of --streams 1 / 2 / 4 on the row-chain build (two half-batch chains on two HIP streams:
# one chain's memory phases under the other's MFMA phases)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4i; mkdir -p $O
cd $R
for i in 1 2; do
  for v in 1 2 4; do
    timeout 300 python bench.py --steps 2 --warmup 1 --streams $v --no-cpu-baseline --no-roofline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams=$v', d['ms_per_step'], d['value'])" | tee -a $O/ab_streams.txt
  done
done

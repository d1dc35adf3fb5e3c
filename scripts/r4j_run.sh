#!/bin/bash
# round 4, GPU call J: do the row-chain launches pay on the batch-2 plans (configs[3] SD-2.1 768^2: 18432 rows at 320 channels)?  TG_RC=0 vs 1
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4j; mkdir -p $O
cd $R
for i in 1 2; do
  for v in 0 1; do
    TG_RC=$v timeout 300 python bench.py --plan sd21 --ddim-steps 10 --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sd21 TG_RC=$v', d['value'], d.get('per_step_ms'))" | tee -a $O/ab_sd21.txt
  done
done

#!/bin/bash
# round 4, GPU call U: norm3 folded into the GEGLU GEMM where it runs on the 128 x 128 kernel (TG_LN_FF_MAX_ROWS 0 / 4096 / 16384): same-box interleaved bench A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4u; mkdir -p $O
cd $R
for i in 1 2 3; do
  for v in 0 4096 16384; do
    TG_LN_FF_MAX_ROWS=$v timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TG_LN_FF_MAX_ROWS=$v', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
  done
done

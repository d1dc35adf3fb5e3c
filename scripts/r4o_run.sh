#!/bin/bash
# round 4, GPU call O: same-box interleaved bench A/B after the rc_front spill fix + global (not FLAT) accesses: TG_RC_MODE 0 (no row-chain), 7 (without rc_front), 15 (default)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4o; mkdir -p $O
cd $R
for i in 1 2; do
  for v in 0 7 15; do
    TG_RC_MODE=$v timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TG_RC_MODE=$v', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
  done
done

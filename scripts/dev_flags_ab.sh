#!/bin/bash
# dev helper: A/B of TG_GEMM_FLAGS experiment variants on ONE GPU box, interleaved (box-to-box variance ~5 %).
#   scripts/dev_flags_ab.sh "0 0x401 0x801 0x2" [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
VARS=${1:-"0"}
ROUNDS=${2:-2}
for i in $(seq $ROUNDS); do
  for v in $VARS; do
    TG_GEMM_FLAGS=$v timeout 500 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags=$v', d['value'], d['ms_per_step'])"
  done
done

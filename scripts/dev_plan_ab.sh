#!/bin/bash
# dev helper: A/B of TG_GEMM_FLAGS values on the configs[3] / configs[4] bench lines, interleaved on ONE box:
#   scripts/dev_plan_ab.sh "0 1024" [rounds] [VAR]      (VAR defaults to TG_GEMM_FLAGS)
R=${GRAFT_REPO_ROOT:-/root/repo}
VALS=$1; ROUNDS=${2:-2}; VAR=${3:-TG_GEMM_FLAGS}
for i in $(seq $ROUNDS); do
  for v in $VALS; do
    export $VAR=$v
    timeout 300 python $R/bench.py --plan sd21 --ddim-steps 10 --steps 2 --warmup 1 > /tmp/ab.out 2> /tmp/ab.err
    tail -1 /tmp/ab.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sd21 $VAR=$v s/step', d['value'], 'unet eager ms', d['per_step_ms']['unet'], 'frac', d['roofline']['frac'], 'guid', d['guidance']['ms'], d['guidance'].get('ms_in_graph'))" 2>/dev/null || { echo "sd21 $VAR=$v FAILED"; tail -5 /tmp/ab.err; }
    timeout 300 python $R/bench.py --plan sdxl --dtype fp16 --ddim-steps 6 --steps 2 --warmup 1 > /tmp/ab.out 2> /tmp/ab.err
    tail -1 /tmp/ab.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sdxl $VAR=$v s/step', d['value'], 'frac', d['roofline']['frac'])" 2>/dev/null || { echo "sdxl $VAR=$v FAILED"; tail -5 /tmp/ab.err; }
  done
done

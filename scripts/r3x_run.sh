#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in 1 2; do
  for f in 0 256; do
    TG_T3_MAX=$f timeout 300 python bench.py --plan sd21 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sd21 T3_MAX=$f', d['value'], d.get('per_step_ms'), d['roofline']['frac'])"
    TG_T3_MAX=$f timeout 300 python bench.py --plan sdxl --dtype fp16 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sdxl T3_MAX=$f', d['value'], d['roofline']['frac'])"
  done
done

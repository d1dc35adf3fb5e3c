#!/bin/bash
# dev helper: A/B two builds of the library on ONE GPU box (box-to-box variance is ~5 %):
#   theatergen_amd/lib/base.so (copy of the previous build) vs the current libtheatergen_hip.so, interleaved.
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for v in base new; do
    if [ $v = base ]; then export THEATERGEN_HIP_LIB=$R/theatergen_amd/lib/base.so; else unset THEATERGEN_HIP_LIB; fi
    timeout 500 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
  done
done

#!/bin/bash
# dev helper: rocprofv3 kernel stats of the graph-replayed bench for TG_GEMM_FLAGS variants + clock / power samples of un-profiled runs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in $1; do
  ( while true; do rocm-smi --showclocks --showpower --json 2>/dev/null | head -c 2000; echo; sleep 0.5; done ) > $R/gpurun_out/smi_$v.log 2>&1 &
  SMI=$!
  TG_GEMM_FLAGS=$v timeout 600 python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $R/gpurun_out/ab_$v.json
  kill $SMI
  TG_GEMM_FLAGS=$v timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$v -- \
    python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_$v.log 2>&1
done

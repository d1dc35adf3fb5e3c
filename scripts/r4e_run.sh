#!/bin/bash
# round 4, GPU call E: row-chain kernels — their tests, the full-size UNet parity tests on the new path, same-box bench A/B (TG_RC=0 vs 1)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4e; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_rowchain_gpu.py -x -q 2>&1 | tail -8 > $O/tests_rowchain.txt; cat $O/tests_rowchain.txt
timeout 1200 python -m pytest tests/test_parity_fullsize_gpu.py tests/test_hotpath_gpu.py -x -q -k "not 50_step" 2>&1 | tail -6 > $O/tests_full.txt; cat $O/tests_full.txt
for i in 1 2; do
  for v in 0 1; do
    TG_RC=$v timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TG_RC=$v', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
  done
done

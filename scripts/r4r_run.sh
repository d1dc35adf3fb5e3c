#!/bin/bash
# round 4, GPU call R: the engine's per-schedule time-projection table: tests + same-box interleaved bench A/B (TG_TPROJ_TABLE 0 / 1)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4r; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_round4_engine_gpu.py tests/test_hotpath_gpu.py tests/test_round3_gpu.py -q -x -m gpu -k "engine or denoise or time_projection or concurrent or loop" 2>&1 | tail -5
for i in 1 2 3; do
  for v in 0 1; do
    TG_TPROJ_TABLE=$v timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TG_TPROJ_TABLE=$v', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
  done
done

#!/bin/bash
# round-3 second pass: GPU tests, configs[3] / [4] A/B of the generic-patch slab rule (TG_GEMM_FLAGS bit 11 = off), headline A/B old vs new library
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3u; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for i in 1 2; do
  for f in 2048 0; do
    TG_GEMM_FLAGS=$f timeout 300 python bench.py --plan sd21 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sd21 flags=$f', d['value'], d.get('per_step_ms'), d['roofline']['frac'])"
    TG_GEMM_FLAGS=$f timeout 300 python bench.py --plan sdxl --dtype fp16 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sdxl flags=$f', d['value'], d['roofline']['frac'])"
  done
done
TG_DUMP_RECS=$O/recs_sd21.json python bench.py --plan sd21 --steps 3 --warmup 1 --no-cpu-baseline > $O/sd21.log 2>&1; python scripts/dev_recs_table.py $O/recs_sd21.json > $O/tab_sd21.txt
TG_DUMP_RECS=$O/recs_sdxl.json python bench.py --plan sdxl --dtype fp16 --steps 3 --warmup 1 --no-cpu-baseline > $O/sdxl.log 2>&1; python scripts/dev_recs_table.py $O/recs_sdxl.json > $O/tab_sdxl.txt
bash scripts/dev_env_ab.sh THEATERGEN_HIP_LIB "$R/theatergen_amd/lib/libtheatergen_hip_old.so -" 3

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3y; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "slab or conv" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
bash scripts/dev_env_ab.sh TG_GEMM_FLAGS "8192 0" 3
for f in 8192 0; do
  TG_GEMM_FLAGS=$f timeout 300 python bench.py --plan sd21 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sd21 flags=$f', d['value'], d['roofline']['frac'])"
  TG_GEMM_FLAGS=$f timeout 300 python bench.py --plan sdxl --dtype fp16 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sdxl flags=$f', d['value'], d['roofline']['frac'])"
done

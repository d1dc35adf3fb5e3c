#!/bin/bash
# round 4, GPU call A: boundary tests + old-.so vs new-.so A/B of the attention d = 40 fix
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_round4_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r4a_tests.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -5 >> gpurun_out/r4a_tests.txt
cat gpurun_out/r4a_tests.txt
timeout 1500 python scripts/ab.py --rounds 3 --out gpurun_out/r4a_ab_attn.json \
  --variant r3:lib=theatergen_amd/lib/libtheatergen_hip_r3.so,THEATERGEN_HIP_ABI_COMPAT=301 --variant new --variant pipe:TG_ATTN_PIPE=1 2>&1 | tail -30

#!/bin/bash
# round 5, call P: one-launch GroupNorm coefficients — parity + same-box graph-replay A/B
mkdir -p gpurun_out/r5p
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_kernels_gpu.py tests/test_rowchain_gpu.py -x -q -k "groupnorm or gn or rc_front or slab" > gpurun_out/r5p/tests.log 2>&1
tail -5 gpurun_out/r5p/tests.log
timeout 1500 python scripts/ab.py --rounds 3 --variant two:TG_GN_ONE_LAUNCH=0 --variant one:TG_GN_ONE_LAUNCH=1 --out gpurun_out/r5p/ab.json > gpurun_out/r5p/ab.log 2>&1
tail -12 gpurun_out/r5p/ab.json

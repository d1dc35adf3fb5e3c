#!/bin/bash
# round 5, call Q: arrival-counter forms (GroupNorm coefficients in one launch; K-split work items finishing their own tiles) with device-scope
# accesses instead of fences — parity + same-box graph-replay A/B of each
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5q; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_round5_gpu.py tests/test_rowchain_gpu.py -x -q -k "split or slab or groupnorm or gn or rc_front or halo or conv" > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 2400 python scripts/ab.py --rounds 3 --out $O/ab.json --variant base:TG_GN_ONE_LAUNCH=0,TG_SPLITK_FINISH=0 --variant gn:TG_GN_ONE_LAUNCH=1,TG_SPLITK_FINISH=0 --variant sk:TG_GN_ONE_LAUNCH=0,TG_SPLITK_FINISH=1 --variant both:TG_GN_ONE_LAUNCH=1,TG_SPLITK_FINISH=1 2>&1 | tail -30

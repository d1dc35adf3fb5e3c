#!/bin/bash
# round 5, GPU call K: chunk rotation of the halo / slab convs (dev bits 32 / 64) same-box A/B, then the round's profile evidence (scripts/r5_profiles.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5k; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "conv" > $O/tests_rot_off.log 2>&1; tail -2 $O/tests_rot_off.log
TG_T160=103 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "conv" > $O/tests_rot_on.log 2>&1; tail -2 $O/tests_rot_on.log
timeout 1800 python scripts/ab.py --rounds 2 --out $O/ab_rot.json --variant t7:TG_T160=7 --variant halo_rot:TG_T160=39 --variant slab_rot:TG_T160=71 2>&1 | tail -20
bash scripts/r5_profiles.sh > $O/profiles.log 2>&1; tail -60 $O/profiles.log

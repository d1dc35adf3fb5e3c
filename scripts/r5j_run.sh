#!/bin/bash
# round 5, GPU call J: K rotation of the 128 x 160 launches (L2 channel spread): tests, isolated sweep, same-box A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5j; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_round5_gpu.py -q -k "128x160" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python scripts/dev_t160.py > $O/t160_sweep_rot.txt 2>&1; cut -c1-260 $O/t160_sweep_rot.txt
timeout 2400 python scripts/ab.py --rounds 3 --out $O/ab.json --variant r4:lib=theatergen_amd/lib/libtheatergen_hip_r4.so --variant t7:TG_T160=7 --variant t23:TG_T160=23 2>&1 | tail -24

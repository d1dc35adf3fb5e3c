#!/bin/bash
# round 5, GPU call G: the 8 x 8 level's halo conv on 128 x 160 tiles (one workgroup per CU, deep weight ring): parity, timing, same-box A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5g; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_round5_gpu.py -q -k "conv_halo_8x8 or 128x160" > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 300 python scripts/dev_halo8.py > $O/halo8.txt 2>&1; cat $O/halo8.txt

timeout 2400 python scripts/ab.py --rounds 2 --out $O/ab.json --variant r4:lib=theatergen_amd/lib/libtheatergen_hip_r4.so --variant t7:TG_T160=7 --variant t15:TG_T160=15 2>&1 | tail -24

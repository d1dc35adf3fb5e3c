#!/bin/bash
# round 5, GPU call B: the 128 x 160 tile variants — isolated sweep, parity tests of the GEMM family, same-box graph-replay A/B (r4 library, TG_T160 = 0 / 1 / 2)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
timeout 600 python scripts/dev_t160.py > $O/t160_sweep.txt 2>&1
cat $O/t160_sweep.txt
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q -k "128x160" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" >> $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 1500 python scripts/ab.py --rounds 2 --out $O/ab_t160.json --variant r4:lib=theatergen_amd/lib/libtheatergen_hip_r4.so --variant off:TG_T160=0 --variant t160:TG_T160=1 --variant t160b:TG_T160=2 2>&1 | tail -30

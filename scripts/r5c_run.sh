#!/bin/bash
# round 5, GPU call C: 128 x 160 tiles (plain + LayerNorm-folded): parity, K-split debug, same-box graph-replay A/B vs the round-4 library
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5c; mkdir -p $O; cd $R
timeout 300 python scripts/dev_t160_split_dbg.py > $O/split_dbg.txt 2>&1; cat $O/split_dbg.txt
timeout 900 python -m pytest tests/test_round5_gpu.py -q -k "128x160" > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "gemm" >> $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 1500 python scripts/ab.py --rounds 3 --out $O/ab_t160.json --variant r4:lib=theatergen_amd/lib/libtheatergen_hip_r4.so --variant off:TG_T160=0 --variant t160:TG_T160=1 2>&1 | tail -30

#!/bin/bash
# round 5, GPU call L: fused norm2 + to_q + cross-attention (tg_xq_attn): parity, timing, same-box A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5l; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_round5_gpu.py -q -k "xq or inner_level" > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout 300 python scripts/dev_xq.py > $O/xq.txt 2>&1; cat $O/xq.txt
timeout 900 python -m pytest tests/test_hotpath_gpu.py -q -x -k "sd15_full or other_baseline or denoise_engine" >> $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 2400 python scripts/ab.py --rounds 2 --out $O/ab.json --variant off:TG_XQ=0 --variant on:TG_XQ=1 2>&1 | tail -16

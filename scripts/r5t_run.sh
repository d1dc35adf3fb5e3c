#!/bin/bash
# round 5, call T: recompute-based self-attention reverse pass (tg_attention_bwd) — parity, reverse-pass suite, timing against the materialised path
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5t; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q -k "reverse_pass or conv_in_on or conv_out_on" > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 1500 python -m pytest tests/test_backward_gpu.py -x -q > $O/tests_bwd.log 2>&1; tail -4 $O/tests_bwd.log
timeout 900 python scripts/dev_attn_bwd.py > $O/timing.txt 2>&1; cat $O/timing.txt

#!/bin/bash
# round 5, GPU call I: operand-pitch experiment (L2 channel spread), then the whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5i; mkdir -p $O; cd $R
timeout 600 python scripts/dev_pitch.py > $O/pitch.txt 2>&1; cat $O/pitch.txt
timeout 2400 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; tail -5 $O/gpu_tests.txt

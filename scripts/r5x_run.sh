#!/bin/bash
# round 5, call X: conv_out staged by eight waves — parity + boundary-conv timing
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5x; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_round5_gpu.py tests/test_kernels_gpu.py -x -q -k "conv_out or conv_in or boundary_convs" > $O/tests.log 2>&1; tail -3 $O/tests.log
python scripts/dev_boundary.py > $O/boundary.txt 2>&1; grep -v amdgpu $O/boundary.txt

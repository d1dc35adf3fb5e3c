#!/bin/bash
# round 5, GPU call H: the whole GPU suite on the current build + the 8 x 8 conv with L2-warm (NC=1) vs rotating (NC=4) weights
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5h; mkdir -p $O; cd $R
for nc in 1 4; do NC=$nc timeout 300 python scripts/dev_halo8.py > $O/halo8_nc$nc.txt 2>&1; cat $O/halo8_nc$nc.txt; done
timeout 2400 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; tail -5 $O/gpu_tests.txt

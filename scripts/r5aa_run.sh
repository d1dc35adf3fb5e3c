#!/bin/bash
# round 5, call AA: gn_partial with 16 loads in flight per lane — parity (bit-identical statistics) + timing + same-box A/B against the call-Y library
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5aa; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_round5_gpu.py -x -q -k "groupnorm or slab_groupnorm or conv_out_on" > $O/tests.log 2>&1; tail -3 $O/tests.log
python scripts/dev_boundary.py 2>&1 | grep -v amdgpu | tee $O/boundary.txt
timeout 1500 python scripts/ab.py --rounds 3 --out $O/ab.json --variant y:lib=theatergen_amd/lib/libtheatergen_hip_y.so --variant new 2>&1 | tail -14

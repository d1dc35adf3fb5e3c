#!/bin/bash
# round 5, call R: conv_in / conv_out on the matrix cores — parity, then same-box graph-replay A/B (old kernels via the dev knobs)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5r; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_kernels_gpu.py -x -q -k "conv_in or conv_out or boundary_convs or groupnorm" > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 1500 python -m pytest tests/test_hotpath_gpu.py tests/test_parity_fullsize_gpu.py -x -q -k "unet or engine or loop or golden" > $O/tests2.log 2>&1; tail -4 $O/tests2.log
timeout 2400 python scripts/ab.py --rounds 3 --out $O/ab.json --variant old:TG_CONV_IN_MFMA=0,TG_CONV_OUT_MFMA=0,TG_CONV_OUT_GN=0 --variant cin:TG_CONV_IN_MFMA=1,TG_CONV_OUT_MFMA=0,TG_CONV_OUT_GN=0 --variant new:TG_CONV_IN_MFMA=1,TG_CONV_OUT_MFMA=1,TG_CONV_OUT_GN=1 2>&1 | tail -24

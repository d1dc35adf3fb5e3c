#!/bin/bash
# round 5, call O: K-group tiles — parity test + isolated sweep
mkdir -p gpurun_out/r5o
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q -k "k_group or 128x160_tiles" > gpurun_out/r5o/tests.log 2>&1
tail -5 gpurun_out/r5o/tests.log
timeout 600 python scripts/dev_kg160.py > gpurun_out/r5o/sweep.txt 2>&1
cat gpurun_out/r5o/sweep.txt

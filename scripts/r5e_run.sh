#!/bin/bash
# round 5, GPU call E: LayerNorm-folded 128 x 160 instances with up-front fragment reads, one-per-CU rounds in the planner: tests, per-shape table, same-box A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5e; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_round5_gpu.py -q -k "128x160" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "gemm or layernorm" >> $O/tests.log 2>&1; tail -3 $O/tests.log
TG_DUMP_RECS=$O/recs.json timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $O/bench.json 2> $O/bench.err
python scripts/dev_recs_table.py $O/recs.json > $O/per_shape.txt 2>&1; grep "plain+ln" $O/per_shape.txt
timeout 2400 python scripts/ab.py --rounds 2 --out $O/ab.json --variant r4:lib=theatergen_amd/lib/libtheatergen_hip_r4.so --variant plain:TG_T160=5 --variant all:TG_T160=7 2>&1 | tail -24

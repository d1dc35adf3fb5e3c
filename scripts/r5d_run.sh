#!/bin/bash
# round 5, GPU call D: 128 x 160 tiles with up-front fragment reads — sweep, tests, per-shape eager tables with the rule off / on, same-box A/B of the rule's parts
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5d; mkdir -p $O; cd $R
timeout 600 python scripts/dev_t160.py > $O/t160_sweep.txt 2>&1; cat $O/t160_sweep.txt
timeout 900 python -m pytest tests/test_round5_gpu.py -q -k "128x160" > $O/tests.log 2>&1; tail -3 $O/tests.log
for m in 0 7; do
  TG_T160=$m TG_DUMP_RECS=$O/recs_t$m.json timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $O/bench_t$m.json 2> $O/bench_t$m.err
  python scripts/dev_recs_table.py $O/recs_t$m.json > $O/per_shape_t$m.txt 2>&1
done
timeout 2400 python scripts/ab.py --rounds 2 --out $O/ab_t160.json --variant r4:lib=theatergen_amd/lib/libtheatergen_hip_r4.so --variant off:TG_T160=0 --variant plain:TG_T160=1 --variant plain_ln:TG_T160=3 --variant plain_rag:TG_T160=5 --variant all:TG_T160=7 --variant r5c_bigw:lib=theatergen_amd/lib/libtheatergen_hip_r5c.so,TG_T160=1 2>&1 | tail -40

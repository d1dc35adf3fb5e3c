#!/bin/bash
# round 5, GPU call N: padded row pitches of the FeedForward hidden tensor / net.2 weight (lda / ldw): parity, per-shape table, same-box A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5n; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_round5_gpu.py -q -k "padded or xq or inner_level or 128x160" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 1200 python -m pytest tests/test_hotpath_gpu.py tests/test_parity_fullsize_gpu.py -q -x -k "unet or engine or loop" >> $O/tests.log 2>&1; tail -3 $O/tests.log
for m in 0 1; do
  TG_FF_PAD=$m TG_DUMP_RECS=$O/recs_pad$m.json timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $O/bench_pad$m.json 2> $O/bench_pad$m.err
  python scripts/dev_recs_table.py $O/recs_pad$m.json > $O/per_shape_pad$m.txt 2>&1; grep "K=  2560\|K=  5120" $O/per_shape_pad$m.txt | cut -c1-150
done
timeout 2400 python scripts/ab.py --rounds 3 --out $O/ab.json --variant off:TG_FF_PAD=0 --variant on:TG_FF_PAD=1 2>&1 | tail -16

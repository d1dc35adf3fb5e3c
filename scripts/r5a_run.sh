#!/bin/bash
# round 5, GPU call A: the new parity tests, a same-box streams 1 vs 2 A/B of the default bench, the per-shape table of the current build
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5a; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_round5_gpu.py tests/test_skinny_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -5 $O/tests.log
for i in 1 2; do
  for s in 1 2; do
    timeout 400 python bench.py --steps 3 --warmup 1 --streams $s --no-cpu-baseline --no-roofline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $s', d['value'], d['ms_per_step'], d.get('sclk_mhz_median'), d.get('power_w_median'))" | tee -a $O/streams_ab.txt
  done
done
TG_DUMP_RECS=$O/recs.json timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $O/bench_line.json 2> $O/bench.err
python scripts/dev_recs_table.py $O/recs.json > $O/per_shape_eager.txt 2>&1
head -c 1500 $O/bench_line.json; echo
head -70 $O/per_shape_eager.txt

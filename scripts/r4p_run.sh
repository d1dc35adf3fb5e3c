#!/bin/bash
# round 4, GPU call P (final sources of the round: rc_front without spills, global instead of FLAT accesses): counters, step breakdown, kernel stats, default bench line, whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4p; mkdir -p $O
bash $R/scripts/r4_profiles.sh > $O/profiles.log 2>&1
# the bench line below quotes the traffic file: use the one just collected on THIS build (the same files are copied into profiles/ by hand afterwards)
cp $R/gpurun_out/r4prof_out/r4_pmc_traffic.json $R/gpurun_out/r4prof_out/r4_pmc_sq.json $R/profiles/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
TG_DUMP_RECS=$O/recs.json timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- \
  python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-other-configs > $O/bench_trace.log 2>&1
python $R/scripts/step_breakdown.py $O/trace $O/recs.json > $O/step_breakdown.txt 2>&1
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/trace
head -34 $O/step_breakdown.txt
cd $R
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.json
timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/tests.txt
cat $O/tests.txt

#!/bin/bash
# Round-6 profile evidence of the bench command (guide: counters in their own passes, kernel-trace only):
#   kernel stats + kernel trace (-> step breakdown, dominant template), --pmc FETCH_SIZE / WRITE_SIZE -> memory-side traffic per kernel (FETCH_SIZE x2 on gfx950),
#   three SQ passes -> matrix-pipe duty, waits, LDS conflicts; kernel stats of the configs[3] / configs[4] lines.  Summaries -> gpurun_out/r6prof_out/ (copied to profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6prof; P=$R/gpurun_out/r6prof_out; mkdir -p $O $P; cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-other-configs > $O/stats.log 2>&1
TG_DUMP_RECS=$O/recs.json timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-other-configs > $O/trace.log 2>&1
TG_DOMINANT_JSON=$P/r6_dominant_template.json python $R/scripts/step_breakdown.py $O/trace $O/recs.json > $P/r6_step_breakdown.txt 2>&1
BENCH3="python $R/bench.py --steps 1 --warmup 0 --ddim-steps 3 --no-cpu-baseline --no-roofline --no-other-configs"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -- $BENCH3 > $O/$c.log 2>&1
done
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/sq$i -- $BENCH3 > $O/sq$i.log 2>&1 || echo "set $i failed" >> $O/fail.log
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sd21 -- python $R/bench.py --plan sd21 --ddim-steps 10 --steps 1 --warmup 1 > $O/stats_sd21.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sdxl -- python $R/bench.py --plan sdxl --dtype fp16 --ddim-steps 6 --steps 1 --warmup 1 > $O/stats_sdxl.log 2>&1
python $R/scripts/r6_profiles_summary.py $O > $O/summary.log 2>&1
find $O -name "*.csv" -size +2M -delete
find $O -type d -name "*_results*" -prune -o -name "*agent_info*" -delete 2>/dev/null
head -40 $P/r6_step_breakdown.txt; cat $P/r6_dominant_template.json; tail -40 $O/summary.log

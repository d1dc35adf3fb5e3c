#!/bin/bash
# Round-4 counter evidence of the bench command on the row-chain build (guide: counters in their own passes, kernel-trace only):
#   --pmc FETCH_SIZE, --pmc WRITE_SIZE -> memory-side traffic per kernel (FETCH_SIZE x2 on gfx950); three SQ passes -> matrix-pipe duty, waits, LDS conflicts
# Summaries: scripts/r4_profiles_summary.py -> gpurun_out/r4prof_out/ (copied to profiles/ by hand).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4prof; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
BENCH3="python $R/bench.py --steps 1 --warmup 0 --ddim-steps 3 --no-cpu-baseline --no-roofline --no-other-configs"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -- $BENCH3 > $O/$c.log 2>&1
done
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/sq$i -- $BENCH3 > $O/sq$i.log 2>&1 || echo "set $i failed" >> $O/fail.log
done
# BASELINE configs[3] / configs[4]: kernel stats of their bench lines on this build
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sd21 -- python $R/bench.py --plan sd21 --ddim-steps 10 --steps 1 --warmup 1 > $O/stats_sd21.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sdxl -- python $R/bench.py --plan sdxl --dtype fp16 --ddim-steps 6 --steps 1 --warmup 1 > $O/stats_sdxl.log 2>&1
python $R/scripts/r4_profiles_summary.py $O > $O/summary.log 2>&1
find $O -name "*.csv" -size +2M -delete
find $O -type d -name "*_results*" -prune -o -name "*agent_info*" -delete 2>/dev/null
tail -60 $O/summary.log

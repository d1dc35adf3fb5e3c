#!/bin/bash
# Round-3 profile evidence of the bench command on the FINAL build (guide: counters in their own passes, kernel-trace only):
#   1. rocprofv3 --kernel-trace --stats            -> gpurun_out/r3prof/stats/
#   2. --pmc FETCH_SIZE, --pmc WRITE_SIZE           -> HBM-side traffic per kernel (FETCH_SIZE x2 on gfx950)
#   3. --pmc SQ busy / MFMA busy / wait breakdown   -> matrix-pipe utilisation of the final GEMM / halo conv / attention kernels
# Summaries are written by scripts/r3_profiles_summary.py into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3prof; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-other-configs"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $BENCH > $O/stats.log 2>&1
BENCH3="python $R/bench.py --steps 1 --warmup 0 --ddim-steps 3 --no-cpu-baseline --no-roofline --no-other-configs"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -- $BENCH3 > $O/$c.log 2>&1
done
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/sq$i -- $BENCH3 > $O/sq$i.log 2>&1 || echo "set $i failed" >> $O/fail.log
done
# BASELINE configs[3] / configs[4]: kernel stats of their bench lines
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sd21 -- python $R/bench.py --plan sd21 --ddim-steps 10 --steps 1 --warmup 1 > $O/stats_sd21.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sdxl -- python $R/bench.py --plan sdxl --dtype fp16 --ddim-steps 6 --steps 1 --warmup 1 > $O/stats_sdxl.log 2>&1
python $R/scripts/r3_profiles_summary.py $O > $O/summary.log 2>&1

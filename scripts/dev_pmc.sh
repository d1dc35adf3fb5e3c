#!/bin/bash
# dev helper: SQ counter passes over one GEMM shape.  usage: scripts/dev_pmc.sh <kind> <force_tile> <tag>
kind=$1; ft=$2; tag=$3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -- python $R/scripts/dev_gemm_one.py $kind $ft > $R/gpurun_out/pmc_${tag}_$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/pmc_${tag}_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm" not in k and "halo" not in k: continue
        agg[(k[:60], r["Counter_Name"])][0] += float(r["Counter_Value"]); agg[(k[:60], r["Counter_Name"])][1] += 1
    for (k, c), (v, n) in sorted(agg.items()):
        print(f"{k:60s} {c:32s} {v / n:16.0f}  (n={n})")
PY

#!/bin/bash
# dev helper: SQ counters of the level-0 self-attention launch (scripts/dev_attn_self.py), one --pmc pass per set
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_attn; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_TRANS_F32"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python $R/scripts/dev_attn_self.py 2 > $O/p$i.log 2>&1 || echo "set $i failed: $set"
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attention_kernel" not in r["Kernel_Name"]: continue
        a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(agg.items()): print(f"{k:32s} {v / n:18.0f}")
PY

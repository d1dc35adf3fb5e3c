#!/bin/bash
# dev helper: executed instructions per wave of the plain GEMM at K = 64 (prologue + one K-tile + epilogue)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_insts; cd /tmp && export TMPDIR=/tmp
cat > /tmp/one.py <<PY
import os, sys
sys.path.insert(0, "$R")
import torch
from theatergen_amd import ops
M, N, K = [int(v) for v in os.environ.get("MNK", "65536,2560,64").split(",")]
a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = torch.randn(N, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, device="cuda").to(torch.bfloat16)
for _ in range(3): ops.linear(a, w, b)
torch.cuda.synchronize()
PY
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_insts/p$i -- python /tmp/one.py > $R/gpurun_out/pmc_insts/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$R/gpurun_out/pmc_insts/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_glds" not in r["Kernel_Name"]: continue
        a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(agg.items()): print(f"{k:28s} {v / n:16.0f}")
PY

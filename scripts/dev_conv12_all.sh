#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3c12; mkdir -p $O; cd $R
for cfg in "0 0" "1 8" "1 12" "1 16" "4 8" "4 16" "2 8" "2 16" "3 8" "3 16" "1 24" "4 24"; do
  set -- $cfg
  timeout 45 python scripts/dev_conv12_one.py $1 $2 0 >> $O/log.txt 2>&1 || echo "ft=$1 fs=$2 c1=0 FAILED rc=$?" >> $O/log.txt
done
grep -v amdgpu.ids $O/log.txt

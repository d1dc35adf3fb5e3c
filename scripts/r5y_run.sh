#!/bin/bash
# round 5, GPU call Y: the round's evidence on the FINAL sources — whole GPU suite, profile passes (scripts/r5_profiles.sh; their summaries are copied into
# profiles/ of this box's copy so that the bench line reads counters collected on the very library it runs), default bench line, boundary-conv timing
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5y; mkdir -p $O; cd $R
python scripts/dev_boundary.py > $O/boundary.txt 2>&1; TG_CONV_IN_MFMA=0 TG_CONV_OUT_MFMA=0 python scripts/dev_boundary.py >> $O/boundary.txt 2>&1; cat $O/boundary.txt
timeout 2700 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
bash scripts/r5_profiles.sh > $O/profiles.log 2>&1; tail -45 $O/profiles.log
cd $R; cp gpurun_out/r5prof_out/*.json gpurun_out/r5prof_out/*.txt profiles/ 2>/dev/null
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; head -c 2500 $O/bench_line.json; echo

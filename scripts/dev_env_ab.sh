#!/bin/bash
# dev helper: A/B of an environment switch on ONE GPU box, interleaved:  scripts/dev_env_ab.sh VAR "val1 val2" [rounds]   ("-" = unset)
R=${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; VALS=$2; ROUNDS=${3:-3}
for i in $(seq $ROUNDS); do
  for v in $VALS; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    timeout 300 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs > /tmp/ab.out 2> /tmp/ab.err
    tail -1 /tmp/ab.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], d['ms_per_step'])" 2>/dev/null \
      || { echo "$VAR=$v FAILED"; tail -5 /tmp/ab.err; }
  done
done

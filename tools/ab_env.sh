#!/bin/bash
# A/B of one environment switch on the headline bench, alternating runs on one box:  tools/ab_env.sh VAR "v1 v2 ..." [rounds]
VAR=$1; VALS=${2:-"0 1"}; N=${3:-5}
for i in $(seq 1 $N); do
  for x in $VALS; do
    export $VAR=$x
    python bench.py --steps 30 --warmup 3 --skip-cpu-baseline --no-batch-mode 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); s=d['stage_ms']; print('$VAR=$x', round(d['ms_per_step'],3), 'trace', round(s['trace_commit'],3), 'permz', round(s['perm_z'],3), 'zcommit', round(s['z_commit'],3))"
  done
done

#!/bin/bash
# A/B of one environment switch on the headline bench, alternating runs on one box:  tools/ab_env.sh VAR [pairs]
VAR=$1; N=${2:-5}
for i in $(seq 1 $N); do
  for x in 0 1; do
    env_line="$VAR=$x"
    export $VAR=$x
    python bench.py --steps 30 --warmup 3 --skip-cpu-baseline --no-batch-mode 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); s=d['stage_ms']; print('$env_line', round(d['ms_per_step'],3), 'trace', round(s['trace_commit'],3), 'permz', round(s['perm_z'],3), 'zcommit', round(s['z_commit'],3))"
  done
done

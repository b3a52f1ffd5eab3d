for i in 1 2 3 4 5; do
  for x in 0 1; do
    SBN_NTT_XCD=$x python bench.py --steps 30 --warmup 3 --skip-cpu-baseline --no-batch-mode 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('xcd=$x', round(d['ms_per_step'],3), round(d['stage_ms']['trace_commit'],3), round(d['stage_ms']['z_commit'],3))"
  done
done

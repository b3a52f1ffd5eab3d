#!/bin/bash
# Kernel-trace summary of one bench run (run on the GPU box): tools/kt_bench.sh [bench args...] -> top kernels by total time
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_ktb
rocprofv3 --kernel-trace --stats -d $OUT/prof_ktb -o kt --output-format csv -- python3 $ROOT/bench.py --steps 6 --warmup 1 --skip-cpu-baseline --no-batch-mode "$@" > $OUT/kt_bench.json 2> $OUT/kt_bench.err
python3 - "$OUT" <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
f = glob.glob(out + "/prof_ktb/**/*kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("%-64s calls %5s avg %9.1f us  %6.2f%%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
d = json.load(open(out + "/kt_bench.json"))
print("ms_per_step", round(d["ms_per_step"], 2), {k: round(v, 2) for k, v in d["stage_ms"].items() if "absorb" not in k and "split" not in k})
PY
rm -rf $OUT/prof_ktb

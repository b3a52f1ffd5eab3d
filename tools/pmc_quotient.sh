#!/bin/bash
# A/B of the quotient kernel's HBM traffic and time (run on the GPU box): workgroup order (blocks, segment) against the
# XCD-aware order in which the four segments of a 256-point block run back to back on one XCD (SBN_QUOTIENT_SWIZZLE).
#   tools/pmc_quotient.sh [table]  -> gpurun_out/pmc_quotient.txt
# Counter passes (FETCH_SIZE, WRITE_SIZE) and the kernel-trace pass are separate runs (counters serialise kernels).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
TABLE=${1:-g1}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for sw in 0 1; do
  export SBN_QUOTIENT_SWIZZLE=$sw
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $OUT/prof_q_${sw}_$c -o q --output-format csv -- python3 $ROOT/bench.py --table $TABLE --steps 2 --warmup 0 --skip-cpu-baseline --no-batch-mode > $OUT/pmc_q_${sw}_$c.log 2>&1 || echo "pass failed: $sw $c"
  done
  rocprofv3 --kernel-trace --stats -d $OUT/prof_q_${sw}_kt -o q --output-format csv -- python3 $ROOT/bench.py --table $TABLE --steps 4 --warmup 1 --skip-cpu-baseline --no-batch-mode > $OUT/pmc_q_${sw}_kt.log 2>&1 || echo "trace pass failed: $sw"
done
python3 - "$OUT" "$TABLE" <<'PY'
import csv, glob, sys, collections, re
out, table = sys.argv[1], sys.argv[2]
lines = ["quotient kernel A/B on %s: HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KB (gfx950 correction, MI355X_MICROARCH.md), time from a separate --kernel-trace --stats pass" % table]
for sw in (0, 1):
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(list)
        for f in glob.glob(out + "/prof_q_%d_%s/**/*counter_collection.csv" % (sw, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c:
                    acc[re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")].append(float(r["Counter_Value"]))
        vals[c] = {k: sum(v) / len(v) for k, v in acc.items()}
    dur = {}
    for f in glob.glob(out + "/prof_q_%d_kt/**/*kernel_stats.csv" % sw, recursive=True):
        for r in csv.DictReader(open(f)):
            dur[re.sub(r"\(.*", "", r["Name"]).replace("void ", "")] = (float(r["AverageNs"]), int(r["Calls"]))
    for k in sorted(vals["FETCH_SIZE"]):
        if "quotient" in k:
            f_, w_ = vals["FETCH_SIZE"].get(k, 0), vals["WRITE_SIZE"].get(k, 0)
            d = dur.get(k, (float("nan"), 0))
            lines.append("swizzle=%d  %-40s FETCH %10.0f KB  WRITE %9.0f KB  -> %7.3f GB per launch   avg %8.1f us over %d calls" % (sw, k[:40], f_, w_, (2 * f_ + w_) * 1024 / 1e9, d[0] / 1e3, d[1]))
open(out + "/pmc_quotient.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/prof_q_*

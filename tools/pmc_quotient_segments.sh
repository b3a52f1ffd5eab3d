#!/bin/bash
# Per-segment HBM traffic of the quotient kernel (run on the GPU box): one FETCH_SIZE pass per segment mask.
#   tools/pmc_quotient_segments.sh [table] -> gpurun_out/pmc_quotient_segments.txt
# Needs a DIAGNOSTIC build of the library: SBN_DIAG_QUOTIENT_SEGMASK is compiled in only with -DSBN_DIAG
#   (make -C starky_bn254_amd/csrc clean && make -C starky_bn254_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSBN_DIAG");
# a production library ignores the variable (an invalid proof must never come out of a stale export).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
TABLE=${1:-g1}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mask in 1 2 4 8 15; do
  SBN_DIAG_QUOTIENT_SEGMASK=$mask rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_qs_$mask -o q --output-format csv -- python3 $ROOT/tools/quotient_diag.py $TABLE 2 > $OUT/pmc_qs_$mask.log 2>&1 || echo "pass failed: $mask"
  SBN_DIAG_QUOTIENT_SEGMASK=$mask rocprofv3 --kernel-trace --stats -d $OUT/prof_qs_kt_$mask -o q --output-format csv -- python3 $ROOT/tools/quotient_diag.py $TABLE 3 > $OUT/pmc_qs_kt_$mask.log 2>&1 || echo "trace pass failed: $mask"
done
python3 - "$OUT" "$TABLE" <<'PY'
import csv, glob, sys, re
out, table = sys.argv[1], sys.argv[2]
lines = ["quotient kernel on %s, one segment at a time (mask bit s = segment s; 15 = all): HBM read bytes per launch = 2 * FETCH_SIZE KB (gfx950 correction)" % table]
for mask in (1, 2, 4, 8, 15):
    v = []
    for f in glob.glob(out + "/prof_qs_%d/**/*counter_collection.csv" % mask, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE" and "quotient_kernel" in r["Kernel_Name"]:
                v.append(float(r["Counter_Value"]))
    d = None
    for f in glob.glob(out + "/prof_qs_kt_%d/**/*kernel_stats.csv" % mask, recursive=True):
        for r in csv.DictReader(open(f)):
            if "quotient_kernel" in r["Name"]:
                d = float(r["AverageNs"]) / 1e3
    lines.append("mask %2d: %7.3f GB read per launch over %d launches, %8.1f us" % (mask, 2 * sum(v) / max(len(v), 1) * 1024 / 1e9, len(v), d or float("nan")))
open(out + "/pmc_quotient_segments.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/prof_qs_*

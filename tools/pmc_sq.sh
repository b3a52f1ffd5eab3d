#!/bin/bash
# One-off SQ counter passes over one proof (run on the GPU box): where do the sponge kernel's issue slots go?
#   tools/pmc_sq.sh  -> gpurun_out/pmc_sq.txt (per kernel: sum of every counter over the run)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_I8"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/prof_sq$i -o s --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 0 --skip-cpu-baseline --no-batch-mode > $OUT/pmc_sq$i.log 2>&1 || echo "pass $i failed: $set"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/prof_sq*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:40]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open(out + "/pmc_sq.txt", "w") as o:
    for k in sorted(tot, key=lambda k: -tot[k].get("SQ_INSTS_VALU", 0))[:14]:
        o.write(k + "\n")
        for c in sorted(tot[k]): o.write("   %-26s %.4g\n" % (c, tot[k][c]))
print(open(out + "/pmc_sq.txt").read()[:6000])
PY
rm -rf $OUT/prof_sq*

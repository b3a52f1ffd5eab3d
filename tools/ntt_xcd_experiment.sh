#!/bin/bash
# Does a transform pass find the previous pass's output in its XCD's L2 when the workgroups of a column are kept on one XCD
# (SBN_NTT_XCD=1: (columns, tiles) grid order) and the working set is cut to sub-chunks (SBN_NTT_SUB columns)?  Per setting:
# ms per G1 proof (bench.py) and the transforms' PMC bytes per proof (FETCH_SIZE / WRITE_SIZE passes).  Run on the GPU box:
#   tools/ntt_xcd_experiment.sh -> gpurun_out/ntt_xcd_experiment.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/ntt_xcd_experiment.txt
for cfg in "0 0" "1 0" "0 16" "1 16" "1 8" "1 32"; do
  set -- $cfg
  export SBN_NTT_XCD=$1
  if [ "$2" = "0" ]; then unset SBN_NTT_SUB; else export SBN_NTT_SUB=$2; fi
  tag="xcd$1_sub$2"
  python3 $ROOT/bench.py --steps 10 --warmup 2 --skip-cpu-baseline --no-batch-mode > $OUT/nx_$tag.json 2> $OUT/nx_$tag.err || { echo "bench failed: $tag"; exit 1; }
  rocprofv3 --pmc FETCH_SIZE -d $OUT/nx_f_$tag -o f --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 0 --skip-cpu-baseline --no-batch-mode > $OUT/nx_f_$tag.log 2>&1 || { echo "pmc failed: $tag"; exit 1; }
  rocprofv3 --pmc WRITE_SIZE -d $OUT/nx_w_$tag -o w --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 0 --skip-cpu-baseline --no-batch-mode > $OUT/nx_w_$tag.log 2>&1 || { echo "pmc failed: $tag"; exit 1; }
  python3 - "$OUT" "$tag" >> $OUT/ntt_xcd_experiment.txt <<'PY'
import csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
def total(kind, counter):
    tot, proofs = 0.0, 0
    for f in glob.glob(f"{out}/nx_{kind}_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            if "ntt_fast_pass_kernel" in r["Kernel_Name"] or "ntt_pass_kernel" in r["Kernel_Name"]: tot += float(r["Counter_Value"])
            if "quotient_combine_kernel" in r["Kernel_Name"]: proofs += 1
    return tot * 1024, max(proofs, 1)
f, pf = total("f", "FETCH_SIZE"); w, pw = total("w", "WRITE_SIZE")
d = json.load(open(f"{out}/nx_{tag}.json"))
print(f"{tag:12s}  {d['ms_per_step']:7.3f} ms/proof   transforms: read {2 * f / pf / 1e9:6.2f} GB (2 x FETCH_SIZE)  written {w / pw / 1e9:6.2f} GB  total {(2 * f / pf + w / pw) / 1e9:6.2f} GB per proof")
PY
  rm -rf $OUT/nx_f_$tag $OUT/nx_w_$tag
done
cat $OUT/ntt_xcd_experiment.txt

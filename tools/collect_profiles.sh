#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline refers to (run on the GPU box through gpurun):
#   tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>_kernel_stats.csv, <tag>_pmc_summary.json, <tag>_bench.json
# Kernel timing and the two PMC passes are separate runs (counters serialise kernels, so their timings are not used).
set -e
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 10 --warmup 2 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -o kt --output-format csv -- python3 $ROOT/bench.py --steps 6 --warmup 1 --skip-cpu-baseline --no-batch-mode > $OUT/${TAG}_kt.log 2>&1
cp $(find $OUT/prof_kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
python3 $ROOT/tools/kernel_trace_summary.py $OUT/prof_kt > $OUT/${TAG}_kernel_trace_by_grid.txt
rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_f -o f --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 0 --skip-cpu-baseline --no-batch-mode > $OUT/${TAG}_pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_w -o w --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 0 --skip-cpu-baseline --no-batch-mode > $OUT/${TAG}_pmc_w.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU -d $OUT/prof_v -o v --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 0 --skip-cpu-baseline --no-batch-mode > $OUT/${TAG}_pmc_v.log 2>&1
python3 $ROOT/tools/summarize_pmc.py $(find $OUT/prof_f -name "*counter_collection.csv" | head -1) $(find $OUT/prof_w -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_summary.json $(find $OUT/prof_v -name "*counter_collection.csv" | head -1)
rm -rf $OUT/prof_kt $OUT/prof_f $OUT/prof_w $OUT/prof_v
echo collected $TAG

#!/bin/bash
# BASELINE config[2] on ONE GPU with the host-core share of an 8-rank node: bench.py --batch 24 with
# the library's worker pool at 32, 16, 8 threads (the Jacobian chains of the device witness run there) and with the chains on
# the device instead.  Writes one JSON line per setting to gpurun_out/<tag>_batch24_*.json.   usage: tools/host_share_rehearsal.sh <tag>
set -u
TAG=${1:-r3}
OUT=gpurun_out
mkdir -p $OUT
for T in 32 16 8 4; do
  SBN_HOST_THREADS=$T timeout -k 10 300 python bench.py --batch 24 > $OUT/${TAG}_batch24_threads$T.json 2> $OUT/${TAG}_batch24_threads$T.err || exit 1
done
SBN_TRACEGEN_DEVICE_CHAIN=1 SBN_HOST_THREADS=8 timeout -k 10 300 python bench.py --batch 24 > $OUT/${TAG}_batch24_device_chain.json 2> $OUT/${TAG}_batch24_device_chain.err || exit 1
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/${TAG}_batch24_*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], round(d["value"],2), "proofs/s", d["config"].get("host_threads_per_rank"))
PY

#!/bin/bash
# what do the 944 zero-extension moves of a permutation (two v_mov_b32 per Goldilocks multiply) cost?
# Builds tools/microbench/sponge_rate.hip three times on the GPU box -- the library's streams, the streams WITHOUT the moves
# (wrong results, same dependency structure otherwise) and with v_lshrrev_b64 in their place -- and prints the chip's
# permutation rate at 1..4 waves per SIMD for each.   usage: bash tools/ab_sponge_moves.sh > gpurun_out/<tag>.txt
set -e
cd "$(dirname "$0")/.."
for v in "" nomov lshr; do
  d=/tmp/sbn_ab_moves_${v:-base}
  rm -rf $d && mkdir -p $d && cp starky_bn254_amd/csrc/*.cuh starky_bn254_amd/csrc/*.inc starky_bn254_amd/csrc/*.hpp $d/
  SBN_GEN_OUT=$d SBN_GEN_VARIANT=$v python3 tools/gen_poseidon_sbox_asm.py > /dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$d tools/microbench/sponge_rate.hip -o $d/sponge_rate 2> /dev/null
  echo "== variant: ${v:-library streams (two v_mov_b32 per multiply)}  (S-box stream: $(grep -c . $d/poseidon_sbox3_asm.inc) lines for three S-boxes)"
  $d/sponge_rate 12400
done

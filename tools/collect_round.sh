# Collects a round's evidence on a GPU box in one gpurun call:  bash tools/collect_round.sh <tag>   (files gpurun_out/<tag>_*)
# GPU tests, tools/collect_profiles.sh (bench line, kernel trace, PMC passes), G2 / Fq12 / split / batch bench lines, the sponge
# microbenchmark, parity under every A/B switch, 2-rank gloo rehearsals of the three bench modes.
TAG=${1:-r4_v5}
set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gpu_tests.log 2>&1; echo rc=$? >> gpurun_out/${TAG}_gpu_tests.log; tail -3 gpurun_out/${TAG}_gpu_tests.log
bash tools/collect_profiles.sh ${TAG} > gpurun_out/${TAG}_collect.log 2>&1; tail -2 gpurun_out/${TAG}_collect.log
cd $GRAFT_REPO_ROOT
python bench.py --table g2 --steps 10 --warmup 2 --skip-cpu-baseline --no-batch-mode > gpurun_out/${TAG}_bench_g2.json 2> gpurun_out/${TAG}_bench_g2.err
python tools/fq12_512_time.py > gpurun_out/${TAG}_fq12_512_time.txt 2>&1; tail -1 gpurun_out/${TAG}_fq12_512_time.txt
python bench.py --split --table g1 --steps 10 --warmup 2 --skip-cpu-baseline > gpurun_out/${TAG}_bench_split_g1_w1.json 2> gpurun_out/${TAG}_bench_split_g1_w1.err
python bench.py --split --table fq12 --num-io 512 --steps 3 --warmup 1 --skip-cpu-baseline > gpurun_out/${TAG}_bench_split_fq12_512_w1.json 2> gpurun_out/${TAG}_bench_split_fq12_512_w1.err
python bench.py --batch 24 --skip-cpu-baseline > gpurun_out/${TAG}_bench_batch24.json 2> gpurun_out/${TAG}_bench_batch24.err
# config[2] literally, and the host share of a rank of an 8-rank node (two host threads: the curve chains of the device witness run eight
# instances per AVX-512 IFMA register there, or on the device when the CPU lacks IFMA)
python bench.py --batch 256 --seed 1000 --skip-cpu-baseline > gpurun_out/${TAG}_bench_batch256_seed1000.json 2> gpurun_out/${TAG}_bench_batch256_seed1000.err
SBN_HOST_THREADS=2 python bench.py --batch 24 --skip-cpu-baseline > gpurun_out/${TAG}_bench_batch24_threads2.json 2> /dev/null
SBN_HOST_THREADS=2 SBN_NO_AVX512=1 python bench.py --batch 24 --skip-cpu-baseline > gpurun_out/${TAG}_bench_batch24_threads2_no_avx512.json 2> /dev/null
SBN_TRACE_TIMING=1 python bench.py --steps 3 --warmup 1 --skip-cpu-baseline --no-batch-mode 2>&1 > /dev/null | grep "device tracegen" | tail -6 > gpurun_out/${TAG}_tracegen_timing.txt
(cd tools/microbench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../starky_bn254_amd/csrc range_check_phases.hip -o /tmp/range_check_phases 2> /dev/null && /tmp/range_check_phases > ../../gpurun_out/${TAG}_range_check_phases.txt)
(cd tools/microbench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../starky_bn254_amd/csrc sponge_rate.hip -o /tmp/sponge_rate && /tmp/sponge_rate 12400 > ../../gpurun_out/${TAG}_sponge_rate.txt)
# every experiment switch (honoured only with SBN_EXPERIMENTAL=1, csrc/settings.hpp) must give the same proof bytes
for sw in SBN_NTT_FUSED=0 SBN_NTT_STREAMS=2 SBN_NTT_XCD=0 SBN_FAST_NTT=0 SBN_NTT_SUB=16 SBN_MERKLE_FUSE=0 SBN_QUOTIENT_TAIL=2 SBN_RANGE_CHECK=1 SBN_RANGE_ASYNC=1 SBN_PERM_Z=1 SBN_QUOTIENT_LOOKUPS=1; do echo -n "$sw: " >> gpurun_out/${TAG}_switch_parity.txt; env SBN_EXPERIMENTAL=1 $sw python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "commit_matches or g1exp_proof or proof_bit_exact or g1stark or g1exp_device_witness" 2>&1 | tail -1 >> gpurun_out/${TAG}_switch_parity.txt; done
# the pipeline variants of the 2^18-row tables (fused middle pass, transform streams, split 1,024-point pass) against the committed oracle digest of config[4]
for sw in SBN_NTT_FUSED=0 SBN_NTT_STREAMS=1 SBN_NTT_SPLIT1024=0 SBN_NTT_CHUNK=48; do echo -n "2^18 rows, $sw: " >> gpurun_out/${TAG}_switch_parity.txt; env SBN_EXPERIMENTAL=1 $sw python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fq12exp_2pow18" 2>&1 | tail -1 >> gpurun_out/${TAG}_switch_parity.txt; done
cat gpurun_out/${TAG}_switch_parity.txt
python -c "
import json
for f in ['${TAG}_bench','${TAG}_bench_g2','${TAG}_bench_split_g1_w1','${TAG}_bench_split_fq12_512_w1','${TAG}_bench_batch24','${TAG}_bench_batch256_seed1000','${TAG}_bench_batch24_threads2','${TAG}_bench_batch24_threads2_no_avx512']:
    try:
        d=json.load(open('gpurun_out/%s.json'%f)); print(f, d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
"
# 2-rank gloo rehearsal of the three bench modes on one GPU (the N > 1 code paths of bench.py; RCCL needs one device per rank)
for mode in "" "--batch 12" "--split --table g1"; do
  tag=$(echo "default$mode" | tr -d ' -' | tr -c 'a-z0-9\n' '_')
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --same-device --steps 4 --warmup 1 --skip-cpu-baseline $mode > gpurun_out/${TAG}_bench_2rank_gloo_$tag.json 2> gpurun_out/${TAG}_bench_2rank_gloo_$tag.err
  python -c "import json,sys; d=json.load(open('gpurun_out/${TAG}_bench_2rank_gloo_$tag.json')); print('2-rank', '$mode', d['value'], d['n_gpus'])"
done

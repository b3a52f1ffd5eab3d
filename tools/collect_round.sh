set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r3_v2_gpu_tests.log 2>&1; echo rc=$? >> gpurun_out/r3_v2_gpu_tests.log; tail -3 gpurun_out/r3_v2_gpu_tests.log
bash tools/collect_profiles.sh r3_v2 > gpurun_out/r3_v2_collect.log 2>&1; tail -2 gpurun_out/r3_v2_collect.log
cd $GRAFT_REPO_ROOT
python bench.py --table g2 --steps 10 --warmup 2 --skip-cpu-baseline --no-batch-mode > gpurun_out/r3_v2_bench_g2.json 2> gpurun_out/r3_v2_bench_g2.err
python tools/fq12_512_time.py > gpurun_out/r3_v2_fq12_512_time.txt 2>&1; tail -1 gpurun_out/r3_v2_fq12_512_time.txt
python bench.py --split --table g1 --steps 10 --warmup 2 --skip-cpu-baseline > gpurun_out/r3_v2_bench_split_g1_w1.json 2> gpurun_out/r3_v2_bench_split_g1_w1.err
python bench.py --split --table fq12 --num-io 512 --steps 3 --warmup 1 --skip-cpu-baseline > gpurun_out/r3_v2_bench_split_fq12_512_w1.json 2> gpurun_out/r3_v2_bench_split_fq12_512_w1.err
python bench.py --batch 24 --skip-cpu-baseline > gpurun_out/r3_v2_bench_batch24.json 2> gpurun_out/r3_v2_bench_batch24.err
(cd tools/microbench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../starky_bn254_amd/csrc sponge_rate.hip -o /tmp/sponge_rate && /tmp/sponge_rate 12618 > ../../gpurun_out/r3_v2_sponge_rate.txt)
for sw in SBN_NTT_FUSED=0 SBN_NTT_XCD=0 SBN_FAST_NTT=0 SBN_NTT_SUB=16; do echo -n "$sw: " >> gpurun_out/r3_v2_switch_parity.txt; env $sw python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "commit_matches or g1exp_proof or proof_bit_exact or g1stark" 2>&1 | tail -1 >> gpurun_out/r3_v2_switch_parity.txt; done
cat gpurun_out/r3_v2_switch_parity.txt
python -c "
import json
for f in ['r3_v2_bench','r3_v2_bench_g2','r3_v2_bench_split_g1_w1','r3_v2_bench_split_fq12_512_w1','r3_v2_bench_batch24']:
    try:
        d=json.load(open('gpurun_out/%s.json'%f)); print(f, d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
"

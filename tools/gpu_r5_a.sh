#!/bin/bash
# round-5 visit A: new sampler tests, two-rank variants, timeline build, quick bench
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_her.py -m gpu -q -x -rfEs 2>&1 | tail -15 | tee $O/tests_her.log
timeout 1200 python -m pytest tests/test_gpu_two_ranks.py -m gpu -q -rfEs 2>&1 | tail -15 | tee $O/tests_two_ranks.log
python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-profile 2>&1 | tail -1 > $O/bench_quick.json; head -c 400 $O/bench_quick.json; echo
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > $O/bench_driver_form_before.json; head -c 300 $O/bench_driver_form_before.json; echo
RLARM_LIB=$PWD/rl_arm_under_sparse_reward_amd/librlarm_hip_tl.so timeout 300 python tools/ubench/split_timeline.py > $O/split_timeline_b256.txt 2>&1; head -30 $O/split_timeline_b256.txt

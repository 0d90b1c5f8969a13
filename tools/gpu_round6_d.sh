#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r06d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_two_ranks.py -q -x -s -k "peertiles" > $O/two_ranks.log 2>&1; echo "two ranks rc $?"; grep -v "amdgpu.ids\|Buffer_size\|Gloo\|socket.cpp" $O/two_ranks.log | tail -40
timeout 600 python -m pytest tests/test_gpu_her.py -q -k "f32 or other_shapes" > $O/her.log 2>&1; echo "her rc $?"; tail -15 $O/her.log
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -q -k "forced_data or first_contact or time_boxed" > $O/contract.log 2>&1; echo "contract rc $?"; tail -30 $O/contract.log
EPISODES=5000,20000 BATCHES=256,65536,262144,1048576 F32_ROWS=1 python tools/ubench/sample_fused.py 2>&1 | grep '^{'

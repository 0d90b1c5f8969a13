#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_full.log 2>&1; tail -8 gpurun_out/pytest_gpu_full.log | cut -c1-250
for fd in 1 0; do
rm -rf gpurun_out/prof_fd$fd && mkdir -p gpurun_out/prof_fd$fd
RLARM_FUSE_DW=$fd rocprofv3 --kernel-trace --stats -d gpurun_out/prof_fd$fd -o trace -- python bench.py --steps 800 --warmup 80 --no-cpu-baseline --no-profile > gpurun_out/prof_fd$fd/bench.log 2>&1
echo "RLARM_FUSE_DW=$fd"; tail -1 gpurun_out/prof_fd$fd/bench.log | cut -c1-200; python tools/trace_summary.py gpurun_out/prof_fd$fd/trace_results.db | head -8
done

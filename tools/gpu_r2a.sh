#!/bin/bash
# round-2 GPU visit: full GPU suite, driver-style bench line, long bench, 2-rank self-launch rehearsal on one GPU
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_full.log 2>&1; tail -40 gpurun_out/pytest_gpu_full.log | cut -c1-300
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.log 2>&1; tail -1 gpurun_out/bench_driver.log | cut -c1-300
timeout 300 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-profile > gpurun_out/bench_4000.log 2>&1; tail -1 gpurun_out/bench_4000.log | cut -c1-300

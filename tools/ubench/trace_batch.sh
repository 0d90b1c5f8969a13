#!/bin/bash
set -u
export TMPDIR=/tmp; mkdir -p gpurun_out
B=${1:-1024}; K=${2:-4}
rm -rf gpurun_out/prof_t && mkdir -p gpurun_out/prof_t
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_t -o trace -- python bench.py --batch $B --replay-k $K --steps 800 --warmup 80 --no-cpu-baseline --no-profile > gpurun_out/prof_t/bench.log 2>&1
tail -1 gpurun_out/prof_t/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], round(d['ms_per_step']*1e3,2))"; python tools/trace_summary.py gpurun_out/prof_t/trace_results.db | sed -n 2,7p

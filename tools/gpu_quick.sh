#!/bin/bash
# quick GPU visit: update tests + bench (no cpu baseline) + kernel trace summary
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-profile 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH', d['value'], 'tr/s', d['ms_per_step']*1e3, 'us/step')"
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o trace -- python bench.py --steps 800 --warmup 80 --no-cpu-baseline --no-profile > gpurun_out/prof_bench.log 2>&1
python tools/trace_summary.py gpurun_out/prof/trace_results.db

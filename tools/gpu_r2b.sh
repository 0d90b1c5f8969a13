#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
one() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['steps'], d['warmup'], d['value'], round(d['ms_per_step']*1e3,2), d['config'].get('shader_clock_mhz_after_run'))"; }
for i in 1 2 3; do timeout 300 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-profile > gpurun_out/b4000_$i.log 2>&1; one gpurun_out/b4000_$i.log; done
timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-profile > gpurun_out/b400.log 2>&1; one gpurun_out/b400.log
timeout 300 python bench.py --steps 40 --warmup 40 --no-cpu-baseline --no-profile > gpurun_out/b40.log 2>&1; one gpurun_out/b40.log
timeout 300 python bench.py --steps 39 --warmup 40 --no-cpu-baseline --no-profile > gpurun_out/b39.log 2>&1; one gpurun_out/b39.log
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o trace -- python bench.py --steps 800 --warmup 80 --no-cpu-baseline --no-profile > gpurun_out/prof_bench.log 2>&1
python tools/trace_summary.py gpurun_out/prof/trace_results.db

#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "1024 4" "4096 4"; do set -- $cfg; B=$1; K=$2
rm -rf gpurun_out/prof_b$B && mkdir -p gpurun_out/prof_b$B
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b$B -o trace -- python bench.py --batch $B --replay-k $K --steps 400 --warmup 80 --no-cpu-baseline --no-profile > gpurun_out/prof_b$B/bench.log 2>&1
echo "== batch $B k $K"; python tools/trace_summary.py gpurun_out/prof_b$B/trace_results.db | sed -n 2,8p
done
for r in 8 16; do RLARM_SLAB_ROWS=$r timeout 300 python bench.py --batch 1024 --steps 2000 --warmup 200 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch 1024 rows=$r', d['value'], round(d['ms_per_step']*1e3,2))"; done
for r in 8 16; do RLARM_SLAB_ROWS=$r timeout 300 python bench.py --batch 768 --steps 2000 --warmup 200 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch 768 rows=$r', d['value'], round(d['ms_per_step']*1e3,2))"; done

#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_update.py -m gpu -q -x -k "variants or repeated or track_oracle or graph_equals or rccl" 2>&1 | grep -E "passed|failed" | tail -2
for cfg in "1024 4" "768 4" "896 4" "512 8" "256 4"; do set -- $cfg; B=$1; K=$2
for side in 0 2 auto; do
if [ $side = auto ]; then unset RLARM_PLAN_SIDE; else export RLARM_PLAN_SIDE=$side; fi
timeout 300 python bench.py --batch $B --replay-k $K --steps 2000 --warmup 200 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $B side=$side', d['value'], round(d['ms_per_step']*1e3,2))"
done; done

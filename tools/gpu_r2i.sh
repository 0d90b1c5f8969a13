#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_update.py -m gpu -q -x -k "variants or repeated or track_oracle or golden" 2>&1 | grep -E "passed|failed|Error" | tail -3
for cfg in "1024 4" "512 8" "768 4" "2048 4" "4096 4" "256 4"; do set -- $cfg; B=$1; K=$2
for t in 32 64; do
RLARM_DW_TILE=$t timeout 300 python bench.py --batch $B --replay-k $K --steps 2000 --warmup 200 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $B tile=$t', d['value'], round(d['ms_per_step']*1e3,2))"
done; done

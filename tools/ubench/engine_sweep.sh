#!/bin/bash
set -u
export TMPDIR=/tmp

for B in 1024 1536 2048 3072 4096; do for v in "RLARM_ENGINE=slab8" "RLARM_ENGINE=slab32"; do
env $v timeout 300 python bench.py --batch $B --steps 800 --warmup 80 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $B $v', d['value'], round(d['ms_per_step']*1e3,2), d['config']['final_losses'])"
done; done

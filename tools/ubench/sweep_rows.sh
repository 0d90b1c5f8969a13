#!/bin/bash
# ms/step of bench.py for a few batch sizes with 4- and 8-row slabs (picks the height threshold in agent.hip)
for b in 384 512 768; do for r in 4 8; do
  v=$(RLARM_SLAB_ROWS=$r python bench.py --batch $b --steps 2000 --warmup 200 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "batch $b rows $r: $v"
done; done

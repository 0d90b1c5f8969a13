#!/bin/bash
# us/step of the cycle graph for batch x slab height x gather-ahead (picks the thresholds in agent.hip)
for b in ${BATCHES:-512 768 1024 1280 1536 2048}; do for r in ${ROWS:-8 16}; do for a in ${AHEADS:-1 0}; do
  v=$(BATCH=$b RLARM_SLAB_ROWS=$r RLARM_AHEAD=$a python tools/ubench/notorch_cycle.py 2>&1 | grep "n_batches=40")
  echo "batch $b rows $r ahead $a: $v"
done; done; done

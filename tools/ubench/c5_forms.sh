#!/bin/bash
# Config 5 at 8 ranks on one device, both exchange forms, three times, with the host's cores kept busy by spinners: the losses must
# be the same line six times (the host feeder's store order does not depend on thread timing; round 6).
export TMPDIR=/tmp
pids=""
for i in $(seq 1 ${HOGS:-16}); do timeout 900 python -c "while True: pass" & pids="$pids $!"; done
for i in 1 2 3; do
for ph in 2 1; do
RLARM_PEER_PHASES=$ph timeout 600 python bench.py --gpus 8 --batch 512 --replay-k 8 --feeder-episodes 8 --steps 40 --warmup 40 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('phases $ph', c['peer_exchange_form'][:10], c['final_losses'], c['replicas_bit_identical'], d['ms_per_step'])"
done; done
kill $pids 2>/dev/null

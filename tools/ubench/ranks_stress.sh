#!/bin/bash
# N ranks on ONE device: how often does the peer exchange (gate kernels: ranks share the device) survive the first cycles?
# usage: tools/ubench/ranks_stress.sh "4 8" 4      (world sizes, repetitions)  -> one line per run
export RLARM_PEER_TIMEOUT_S=${RLARM_PEER_TIMEOUT_S:-4}
for W in $1; do for PH in 2 1; do for i in $(seq 1 ${2:-3}); do
  out=$(RLARM_BENCH_ALTERNATIVES=0 RLARM_PEER_PHASES=$PH timeout 300 python bench.py --gpus $W --episodes 64 --steps 80 --warmup 40 --no-cpu-baseline --no-profile 2>&1)
  line=$(echo "$out" | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(c['exchange'], (c['peer_exchange_form'] or '-')[:9], c['replicas_bit_identical'], round(d['ms_per_step']*1e3,1), 'us/step')" 2>/dev/null)
  echo "W=$W phases=$PH run=$i: ${line:-FAILED} $(echo "$out" | grep -c 'failed in the first cycle') fallback(s)"
done; done; done

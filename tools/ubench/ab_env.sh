#!/bin/bash
# same-box A/B of two environment settings (e.g. "RLARM_SPLIT=0" vs "RLARM_AB=default"), N alternating runs each:
#   tools/ubench/ab_env.sh "RLARM_SPLIT=0" "RLARM_AB=default" [rounds]      AB_FLAGS="--batch 1024" AB_STEPS=2000 for other shapes
A="$1"; B="$2"; N=${3:-3}
for i in $(seq $N); do for v in "$A" "$B"; do
  echo -n "[$v] "; env $v timeout 300 python bench.py --steps ${AB_STEPS:-4000} --warmup 400 --no-cpu-baseline --no-profile $AB_FLAGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,3), d['config']['final_losses'])"
done; done

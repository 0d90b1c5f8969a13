#!/bin/bash
# same-box A/B of library builds, alternating runs: tools/ubench/ab_libs.sh new old [pad ...]  ("new" = the in-tree library,
# anything else = rl_arm_under_sparse_reward_amd/librlarm_hip_<name>.so); AB_FLAGS="--batch 1024" for other shapes
LIBS=${@:-new old}
for i in 1 2 3; do for l in $LIBS; do L=""; [ "$l" != new ] && L="RLARM_LIB=$PWD/rl_arm_under_sparse_reward_amd/librlarm_hip_$l.so"; echo -n "lib=$l "; env $L timeout 200 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-profile $AB_FLAGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,3), d['config']['final_losses'])"; done; done

#!/bin/bash
# A/B of the weight-gradient kernels: 32 x 32 tiles (gemm_lds.h) vs 64 x 64 split tiles (dw64.h)
set -u
export TMPDIR=/tmp
for B in ${BATCHES:-1024 2048 3072 4096}; do for v in "RLARM_DW64=0" "RLARM_DW64=1" "RLARM_DW64=s2" "RLARM_DW64=s4"; do
env $v timeout 300 python bench.py --batch $B --steps 800 --warmup 80 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $B $v', d['value'], round(d['ms_per_step']*1e3,2), d['config']['final_losses'])"
done; done

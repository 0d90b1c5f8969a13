#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_update.py -m gpu -q -x -k "rccl_path" 2>&1 | tail -5
for t in peer torch; do RLARM_COMM=$t timeout 600 python bench.py --gpus 2 --steps 80 --warmup 40 --no-cpu-baseline > gpurun_out/bench_2rank_$t.log 2>&1; tail -1 gpurun_out/bench_2rank_$t.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['n_gpus'], d['value'], round(d['ms_per_step']*1e3,1), d['config'].get('exchange'), d['config'].get('cycle_mode'))" || tail -5 gpurun_out/bench_2rank_$t.log; done
RLARM_BENCH_FORCE_DP=1 RLARM_COMM=peer timeout 300 python bench.py --steps 800 --warmup 80 --no-cpu-baseline --no-profile 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced dp1 peer', d['value'], round(d['ms_per_step']*1e3,2), d['config'].get('exchange'), d['config'].get('cycle_mode'))"
RLARM_BENCH_FORCE_DP=1 RLARM_COMM=rccl timeout 300 python bench.py --steps 800 --warmup 80 --no-cpu-baseline --no-profile 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced dp1 rccl', d['value'], round(d['ms_per_step']*1e3,2), d['config'].get('exchange'), d['config'].get('cycle_mode'))"

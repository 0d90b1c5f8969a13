#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
show() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], round(d['ms_per_step']*1e3,2), json.dumps(d.get('host_feeder'))[:400])" || tail -5 $1; }
timeout 300 python bench.py --batch 512 --replay-k 8 --steps 20000 --warmup 400 --no-cpu-baseline --no-profile > gpurun_out/c5_nofeeder.log 2>&1; show gpurun_out/c5_nofeeder.log
timeout 300 python bench.py --batch 512 --replay-k 8 --steps 20000 --warmup 400 --no-cpu-baseline --no-profile --feeder-envs 64 --feeder-workers 8 > gpurun_out/c5_feeder64.log 2>&1; show gpurun_out/c5_feeder64.log
timeout 300 python bench.py --steps 20000 --warmup 400 --no-cpu-baseline --no-profile --feeder-envs 64 --feeder-workers 16 > gpurun_out/b256_feeder64.log 2>&1; show gpurun_out/b256_feeder64.log
nproc

#!/bin/bash
# Round-6 visit A (through gpurun): the whole GPU suite with the data-parallel split forms, then the forced data-parallel lines at
# world 1 (peer tile-wise = k_fb_split8<1>; RCCL and the separate exchange launch = k_fb_split8<2>) beside the single-rank line.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06a; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.log
B="python bench.py --steps 4000 --warmup 400 --no-cpu-baseline"
$B > $O/bench_b256.log 2>&1; tail -1 $O/bench_b256.log > $O/bench_n1_b256.json
RLARM_BENCH_FORCE_DP=1 $B > $O/dp.log 2>&1; tail -1 $O/dp.log > $O/bench_n1_b256_forced_dp_world1.json
RLARM_BENCH_FORCE_DP=1 RLARM_PEER_TILES=0 $B > $O/dp0.log 2>&1; tail -1 $O/dp0.log > $O/bench_n1_b256_forced_dp_world1_separate_exchange_launch.json
RLARM_BENCH_FORCE_DP=1 RLARM_COMM=rccl $B > $O/dpr.log 2>&1; tail -1 $O/dpr.log > $O/bench_n1_b256_forced_dp_world1_rccl.json
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read())
    r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], round(d['ms_per_step']*1e3,3), 'us/update', d['config'].get('exchange'), d['config']['engine'].get('kernels_per_update'), r.get('kernel'), r.get('frac'), r.get('avg_launch_us'))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done

for cfg in "480 4 RLARM_SLAB_ROWS=4" "480 4 RLARM_SLAB_ROWS=8" "512 4 RLARM_SLAB_ROWS=4" "512 4 RLARM_SLAB_ROWS=8" "512 8 RLARM_SLAB_ROWS=4" "512 8 RLARM_SLAB_ROWS=8" "448 4 RLARM_SLAB_ROWS=4" "448 4 RLARM_SLAB_ROWS=8" "544 4 RLARM_SLAB_ROWS=4" "544 4 RLARM_SLAB_ROWS=8"; do set -- $cfg
env $3 python bench.py --batch $1 --replay-k $2 --steps 1200 --warmup 120 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $1 k$2 $3', d['value'], round(d['ms_per_step']*1e3,2))"
done

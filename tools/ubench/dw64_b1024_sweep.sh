export TMPDIR=/tmp
for B in 1024; do for v in "RLARM_AB=default" "RLARM_DW64=s2" "RLARM_DW64=s3" "RLARM_DW64=s4" "RLARM_DW64=s6" "RLARM_AB=default"; do
env $v timeout 300 python bench.py --batch $B --steps 2000 --warmup 200 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $B $v', round(d['ms_per_step']*1e3,2), d['config']['final_losses'])"
done; done
for v in "RLARM_AB=default" "RLARM_DW64=s2" "RLARM_DW64=s4"; do
env $v timeout 300 python bench.py --batch 512 --replay-k 8 --steps 2000 --warmup 200 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch 512k8 $v', round(d['ms_per_step']*1e3,2), d['config']['final_losses'])"
done

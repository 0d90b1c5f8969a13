for B in 512 1024; do K=4; [ $B = 512 ] && K=8
  for v in "RLARM_DW64=0" "RLARM_DW64=s1" "RLARM_DW64=s2" "RLARM_DW64=s3" "RLARM_DW64=s4"; do
    echo -n "batch $B k $K $v: "; env $v timeout 200 python bench.py --batch $B --replay-k $K --steps 2000 --warmup 200 --no-cpu-baseline --no-profile 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2))"
  done
done

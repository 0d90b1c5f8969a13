python tools/ubench/dp_path.py cycle 2>&1 | grep "dp path\|cycle mode"
python tools/ubench/dp_path.py eager 2>&1 | grep "dp path\|cycle mode"
RLARM_BENCH_FORCE_DP=1 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-profile 2>&1 | tail -1 | cut -c1-300

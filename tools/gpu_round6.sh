#!/bin/bash
# Round-6 measurement visit (run through gpurun): bench lines, rocprofv3 kernel traces, FETCH/WRITE PMC passes and SQ counters for
# the headline shape; traces + bench lines of the per-GPU shapes of BASELINE configs 3/4/5; the forced data-parallel forms at
# world 1 (peer tile-wise = k_fb_split8<1>, RCCL / separate exchange = k_fb_split8<2>) with traces and traffic; the stand-alone
# samplers (float64 rows and the float32 throughput rows) with traffic; the 8-rank rehearsal line.  Output: gpurun_out/r06/,
# copied into profiles/r06_* by the caller.  Counter passes are separate runs with --kernel-trace only.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
trace() {   # trace <name> <env...> -- <bench flags...>
  local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local CMD="python bench.py $* --no-cpu-baseline --no-profile"
  local d=$O/trace_$name; rm -rf $d; mkdir -p $d
  env "${envs[@]}" rocprofv3 --kernel-trace --stats -d $d -o trace -- $CMD > $d/log.txt 2>&1
  python tools/trace_summary.py $(find $d -name "*.db" | head -1) "${envs[*]} $CMD" > $O/kernel_trace_$name.txt
}
pmc() {     # pmc <name> <env...> -- <bench flags...>
  local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    local d=$O/pmc_${c}_$name; rm -rf $d; mkdir -p $d
    env "${envs[@]}" timeout 300 rocprofv3 --kernel-trace --pmc $c -d $d -o pmc -- python bench.py $* --steps 120 --warmup 40 --no-cpu-baseline --no-profile > $d/log.txt 2>&1
  done
  python tools/pmc_summary.py $(find $O/pmc_FETCH_SIZE_$name -name "*.db" | head -1) $(find $O/pmc_WRITE_SIZE_$name -name "*.db" | head -1) --json $O/pmc_traffic_$name.json --note "$name ${envs[*]} $*" > $O/pmc_fetch_write_$name.txt 2>&1
}
trace b256_k4 X=1 -- --batch 256 --replay-k 4 --steps 2000 --warmup 400
trace b1024_k4 X=1 -- --batch 1024 --replay-k 4 --steps 800 --warmup 80
trace b512_k8 X=1 -- --batch 512 --replay-k 8 --steps 800 --warmup 80
trace b4096_k4 X=1 -- --batch 4096 --replay-k 4 --steps 400 --warmup 80
trace b256_k4_forced_dp_peer RLARM_BENCH_FORCE_DP=1 RLARM_COMM=peer -- --batch 256 --steps 2000 --warmup 400
trace b256_k4_forced_dp_rccl RLARM_BENCH_FORCE_DP=1 RLARM_COMM=rccl -- --batch 256 --steps 2000 --warmup 400
d2=$O/csv_b256; rm -rf $d2; mkdir -p $d2
rocprofv3 --kernel-trace --stats --output-format csv -d $d2 -o trace -- python bench.py --steps 2000 --warmup 400 --no-cpu-baseline --no-profile > $d2/log.txt 2>&1
cp $(find $d2 -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_b256_k4.csv 2>/dev/null
pmc b256 X=1 -- --batch 256
pmc b1024 X=1 -- --batch 1024
pmc b512 X=1 -- --batch 512 --replay-k 8
pmc b256_forced_dp_peer RLARM_BENCH_FORCE_DP=1 RLARM_COMM=peer -- --batch 256
pmc b256_forced_dp_rccl RLARM_BENCH_FORCE_DP=1 RLARM_COMM=rccl -- --batch 256
# samplers: sweep, traces, traffic
{ EPISODES=5000,10000,20000 BATCHES=256,4096,65536,262144,1048576 python tools/ubench/sample_fused.py; EPISODES=5000,10000,20000 BATCHES=256,4096,65536,262144,1048576 F32_ROWS=1 python tools/ubench/sample_fused.py; } 2>&1 | grep '^{' > $O/sample_kernel_shard_sweep.txt
for mode in 0 1; do for c in FETCH_SIZE WRITE_SIZE; do
  d=$O/pmcs_${c}_$mode; rm -rf $d; mkdir -p $d
  F32_ROWS=$mode EPISODES=5000,10000 BATCHES=256,262144 REPS=10 timeout 300 rocprofv3 --kernel-trace --pmc $c -d $d -o pmc -- python tools/ubench/sample_fused.py > $d/log.txt 2>&1
done; done
python - <<'PY'
import sqlite3, glob, json
out = {}
for mode in (0, 1):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        dbs = glob.glob(f"gpurun_out/r06/pmcs_{c}_{mode}/**/*.db", recursive=True)
        if not dbs:
            continue
        cur = sqlite3.connect(dbs[0]).cursor()
        for r in cur.execute(f"select kernel_name, grid_size_x, count(*), avg(value) from counters_collection where counter_name='{c}' and kernel_name like '%gather_%' group by kernel_name, grid_size_x"):
            out.setdefault(f"{r[0].split('(')[0]}|grid{r[1]}", {})[c] = r[3]
json.dump(out, open("gpurun_out/r06/pmc_sample_raw.json", "w"), indent=1)
print(json.dumps(out))
PY
d=$O/trace_sample; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --stats -d $d -o t -- env EPISODES=5000 BATCHES=256,262144 REPS=20 python tools/ubench/sample_fused.py > $d/log.txt 2>&1
python tools/trace_summary.py $(find $d -name "*.db" | head -1) "EPISODES=5000 BATCHES=256,262144 REPS=20 python tools/ubench/sample_fused.py" > $O/kernel_trace_sample_fused.txt
d=$O/trace_sample32; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --stats -d $d -o t -- env F32_ROWS=1 EPISODES=5000 BATCHES=256,262144 REPS=20 python tools/ubench/sample_fused.py > $d/log.txt 2>&1
python tools/trace_summary.py $(find $d -name "*.db" | head -1) "F32_ROWS=1 EPISODES=5000 BATCHES=256,262144 REPS=20 python tools/ubench/sample_fused.py" > $O/kernel_trace_sample_f32_rows.txt
# the bench lines come AFTER the counter passes and the traces: roofline.traffic / the committed kernel averages are read from profiles/r06_*
for n in b256 b1024 b512 b256_forced_dp_peer b256_forced_dp_rccl; do cp $O/pmc_traffic_$n.json profiles/r06_pmc_traffic_$n.json; done
for n in b256_k4 b1024_k4 b512_k8 b4096_k4 b256_k4_forced_dp_peer b256_k4_forced_dp_rccl; do cp $O/kernel_trace_$n.txt profiles/r06_kernel_trace_$n.txt; done
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.log 2>&1; tail -1 $O/bench_driver_style.log > $O/bench_n1_b256_driver_style.json
python bench.py --steps 4000 --warmup 400 > $O/bench_b256.log 2>&1; tail -1 $O/bench_b256.log > $O/bench_n1_b256.json
RLARM_SPLIT=0 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline > $O/bench_b256_two_launch.log 2>&1; tail -1 $O/bench_b256_two_launch.log > $O/bench_n1_b256_two_launch_form.json
RLARM_BENCH_FORCE_DP=1 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline > $O/dp.log 2>&1; tail -1 $O/dp.log > $O/bench_n1_b256_forced_dp_world1.json
RLARM_BENCH_FORCE_DP=1 RLARM_PEER_TILES=0 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline > $O/dp0.log 2>&1; tail -1 $O/dp0.log > $O/bench_n1_b256_forced_dp_world1_separate_exchange_launch.json
RLARM_BENCH_FORCE_DP=1 RLARM_COMM=rccl python bench.py --steps 4000 --warmup 400 --no-cpu-baseline > $O/dpr.log 2>&1; tail -1 $O/dpr.log > $O/bench_n1_b256_forced_dp_world1_rccl.json
RLARM_BENCH_FORCE_DP=1 RLARM_SPLIT=0 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-profile > $O/dps0.log 2>&1; tail -1 $O/dps0.log > $O/bench_n1_b256_forced_dp_world1_two_launch_form.json
RLARM_BENCH_FORCE_DP=1 RLARM_COMM=rccl RLARM_SPLIT=0 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-profile > $O/dprs0.log 2>&1; tail -1 $O/dprs0.log > $O/bench_n1_b256_forced_dp_world1_rccl_two_launch_form.json
python bench.py --batch 1024 --steps 2000 --warmup 200 --cpu-seconds 10 > $O/bench_b1024.log 2>&1; tail -1 $O/bench_b1024.log > $O/bench_n1_b1024.json
python bench.py --batch 512 --replay-k 8 --steps 2000 --warmup 200 --cpu-seconds 10 > $O/bench_b512k8.log 2>&1; tail -1 $O/bench_b512k8.log > $O/bench_n1_b512_k8.json
python bench.py --batch 4096 --steps 800 --warmup 80 --no-cpu-baseline > $O/bench_b4096.log 2>&1; tail -1 $O/bench_b4096.log > $O/bench_n1_b4096.json
# 8 ranks on the one device: the line a multi-GPU run prints, exchange_alternatives (time-boxed) and device identity included (rehearsal, never a number to quote)
timeout 900 python bench.py --gpus 8 --episodes 64 --steps 80 --warmup 40 --no-cpu-baseline --no-profile > $O/bench_rehearsal_8ranks.log 2>&1; grep '^{"metric"' $O/bench_rehearsal_8ranks.log > $O/bench_rehearsal_8_ranks_one_device.json
timeout 600 python bench.py --gpus 2 --episodes 64 --steps 80 --warmup 40 --no-cpu-baseline > $O/bench_rehearsal_2ranks.log 2>&1; grep '^{"metric"' $O/bench_rehearsal_2ranks.log > $O/bench_rehearsal_2_ranks_one_device.json
i=0; rm -f $O/pmc_sq_b256.txt
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY"; do
  i=$((i+1)); d=$O/pmcq_$i; rm -rf $d; mkdir -p $d
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $d -o pmc -- python bench.py --steps 120 --warmup 40 --no-cpu-baseline --no-profile > $d/log.txt 2>&1
  python tools/pmc_summary.py $(find $d -name "*.db" | head -1) 2>&1 | grep "k_fb_\|k_gemm_lds" >> $O/pmc_sq_b256.txt
done
timeout 300 python tools/ubench/level1_gpu.py 2>&1 | grep -v "amdgpu.ids\|Buffer_size" > $O/level1_gpu_b256.txt
rm -rf $O/trace_* $O/csv_b* $O/pmc_FETCH* $O/pmc_WRITE* $O/pmcs_* $O/pmcq_*
ls $O; head -8 $O/kernel_trace_b256_k4.txt; head -8 $O/kernel_trace_b256_k4_forced_dp_peer.txt; cat $O/pmc_fetch_write_b256.txt | grep "k_fb\|k_gemm"; for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], round(d['ms_per_step']*1e3,3), 'us/update', d['config'].get('exchange'), d['config']['engine'].get('kernels_per_update'), r.get('kernel'), r.get('frac'), r.get('avg_launch_us'), r.get('duration_source'))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done

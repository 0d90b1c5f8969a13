#!/bin/bash
# Round-5 measurement visit (run through gpurun): bench lines, rocprofv3 kernel traces (+ the tool's own --stats CSV), FETCH/WRITE
# PMC passes, L2 (TCC) counters and SQ counters for the headline shape and the per-GPU shapes of BASELINE configs 3/4/5, the split
# launch's per-workgroup time line, the forced data-parallel lines, the stand-alone samplers.  Output: gpurun_out/r05/, copied into
# profiles/r05_* by the caller.  Counter passes are separate runs with --kernel-trace only (no --stats / sys-trace).
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for cfg in "256 4 2000 400" "1024 4 800 80" "512 8 800 80" "4096 4 400 80"; do set -- $cfg; B=$1; K=$2; S=$3; W=$4
  CMD="python bench.py --batch $B --replay-k $K --steps $S --warmup $W --no-cpu-baseline --no-profile"
  d=$O/trace_b$B; rm -rf $d; mkdir -p $d
  rocprofv3 --kernel-trace --stats -d $d -o trace -- $CMD > $d/log.txt 2>&1
  python tools/trace_summary.py $d/trace_results.db "$CMD" > $O/kernel_trace_b${B}_k$K.txt
  d2=$O/csv_b$B; rm -rf $d2; mkdir -p $d2
  rocprofv3 --kernel-trace --stats --output-format csv -d $d2 -o trace -- $CMD > $d2/log.txt 2>&1
  cp $d2/trace_kernel_stats.csv $O/rocprofv3_kernel_stats_b${B}_k$K.csv 2>/dev/null || ls $d2
done
for cfg in "256 4" "512 8" "1024 4" "4096 4"; do set -- $cfg; B=$1; K=$2
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$O/pmc_${c}_b$B; rm -rf $d; mkdir -p $d
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d $d -o pmc -- python bench.py --batch $B --replay-k $K --steps 120 --warmup 40 --no-cpu-baseline --no-profile > $d/log.txt 2>&1
  done
  python tools/pmc_summary.py $O/pmc_FETCH_SIZE_b$B/pmc_results.db $O/pmc_WRITE_SIZE_b$B/pmc_results.db --json $O/pmc_traffic_b$B.json --note "batch $B replay_k $K" > $O/pmc_fetch_write_b$B.txt 2>&1
done
for B in 256 1024; do BATCH=$B tools/ubench/pmc_tcc.sh > $O/pmc_tcc_b$B.txt 2>&1; done
# the bench lines come AFTER the counter passes and the traces: roofline.traffic / the committed kernel averages are read from profiles/r05_*
# and checked against the kernel sources' fingerprint (traffic_stale, duration_stale)
for B in 256 512 1024 4096; do cp $O/pmc_traffic_b$B.json profiles/r05_pmc_traffic_b$B.json; done
for cfg in "256 4" "512 8" "1024 4" "4096 4"; do set -- $cfg; cp $O/kernel_trace_b$1_k$2.txt profiles/r05_kernel_trace_b$1_k$2.txt; done
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.log 2>&1; tail -1 $O/bench_driver_style.log > $O/bench_n1_b256_driver_style.json
python bench.py --steps 4000 --warmup 400 > $O/bench_b256.log 2>&1; tail -1 $O/bench_b256.log > $O/bench_n1_b256.json
RLARM_SPLIT=0 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline > $O/bench_b256_two_launch.log 2>&1; tail -1 $O/bench_b256_two_launch.log > $O/bench_n1_b256_two_launch_form.json
RLARM_BENCH_FORCE_DP=1 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-profile > $O/bench_b256_dp.log 2>&1; tail -1 $O/bench_b256_dp.log > $O/bench_n1_b256_forced_dp_world1.json
RLARM_BENCH_FORCE_DP=1 RLARM_PEER_TILES=0 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-profile > $O/bench_b256_dp0.log 2>&1; tail -1 $O/bench_b256_dp0.log > $O/bench_n1_b256_forced_dp_world1_separate_exchange_launch.json
RLARM_BENCH_FORCE_DP=1 RLARM_COMM=rccl python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-profile > $O/bench_b256_dp_rccl.log 2>&1; tail -1 $O/bench_b256_dp_rccl.log > $O/bench_n1_b256_forced_dp_world1_rccl.json
python bench.py --batch 1024 --steps 2000 --warmup 200 --cpu-seconds 10 > $O/bench_b1024.log 2>&1; tail -1 $O/bench_b1024.log > $O/bench_n1_b1024.json
python bench.py --batch 512 --replay-k 8 --steps 2000 --warmup 200 --cpu-seconds 10 > $O/bench_b512k8.log 2>&1; tail -1 $O/bench_b512k8.log > $O/bench_n1_b512_k8.json
python bench.py --batch 4096 --steps 800 --warmup 80 --no-cpu-baseline > $O/bench_b4096.log 2>&1; tail -1 $O/bench_b4096.log > $O/bench_n1_b4096.json
# 8 ranks on the one device: the line a multi-GPU run prints, exchange_alternatives and device identity included (rehearsal, never a number to quote)
timeout 900 python bench.py --gpus 8 --episodes 64 --steps 80 --warmup 40 --no-cpu-baseline --no-profile > $O/bench_rehearsal_8ranks.log 2>&1; grep '^{"metric"' $O/bench_rehearsal_8ranks.log > $O/bench_rehearsal_8_ranks_one_device.json
i=0; rm -f $O/pmc_sq_b256.txt
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY"; do
  i=$((i+1)); d=$O/pmcs_$i; rm -rf $d; mkdir -p $d
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $d -o pmc -- python bench.py --steps 120 --warmup 40 --no-cpu-baseline --no-profile > $d/log.txt 2>&1
  python tools/pmc_summary.py $d/pmc_results.db 2>&1 | grep "k_fb_\|k_gemm_lds" >> $O/pmc_sq_b256.txt
done
RLARM_LIB=$PWD/rl_arm_under_sparse_reward_amd/librlarm_hip_tl.so timeout 300 python tools/ubench/split_timeline.py > $O/split_timeline_b256.txt 2>&1
tools/ubench/ab_env.sh "RLARM_SPLIT=0" "RLARM_AB=default" 3 > $O/ab_split_b256.txt 2>&1
timeout 300 python tools/ubench/level1_gpu.py 2>&1 | grep -v "amdgpu.ids\|Buffer_size" > $O/level1_gpu_b256.txt
rm -rf $O/trace_b* $O/csv_b* $O/pmc_FETCH* $O/pmc_WRITE* $O/pmcs_* gpurun_out/pmct_*
ls $O; cat $O/kernel_trace_b256_k4.txt | head -8; cat $O/pmc_fetch_write_b256.txt | grep "k_fb\|k_gemm"; head -c 600 $O/bench_n1_b256.json

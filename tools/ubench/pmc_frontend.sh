#!/bin/bash
# Instruction-cache, scalar-cache and address-translation counters per kernel (separate passes; kernel-trace only): what do the first
# workgroups of a launch wait for?  (profiles/r05_ab_actor_tile_launch_tail.txt)
set -u
export TMPDIR=/tmp
B=${BATCH:-256}; K=${REPLAY_K:-4}
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_STALL"; do
  i=$((i+1))
  d=gpurun_out/pmcf_$i; rm -rf $d; mkdir -p $d
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $d -o pmc -- python bench.py --batch $B --replay-k $K --steps 120 --warmup 40 --no-cpu-baseline --no-profile > $d/log.txt 2>&1
  python tools/pmc_summary.py $d/pmc_results.db 2>&1 | grep "k_fb_\|k_gemm_lds\|k_dw64" | head -10
done
rm -rf gpurun_out/pmcf_*

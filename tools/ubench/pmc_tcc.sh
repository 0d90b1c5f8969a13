#!/bin/bash
# L2 (TCC) request / hit / miss / fabric-read counters per kernel at one batch size: are the FETCH_SIZE bytes of the
# weight-gradient launch real L2 misses?  Separate passes; kernel-trace only.
set -u
export TMPDIR=/tmp
B=${BATCH:-1024}; K=${REPLAY_K:-4}
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  d=gpurun_out/pmct_$i; rm -rf $d; mkdir -p $d
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $d -o pmc -- python bench.py --batch $B --replay-k $K --steps 120 --warmup 40 --no-cpu-baseline --no-profile > $d/log.txt 2>&1
  python tools/pmc_summary.py $d/pmc_results.db 2>&1 | grep "k_fb_\|k_gemm_lds\|k_dw64" | head -8
done

#!/bin/bash
# issue-side counters of the two hot kernels (separate passes; kernel-trace only): matrix-pipe busy cycles, LDS bank
# conflicts, wait cycles
set -u
export TMPDIR=/tmp
B=${BATCH:-256}
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  d=gpurun_out/pmcs_$i; rm -rf $d; mkdir -p $d
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $d -o pmc -- python bench.py --batch $B --steps 120 --warmup 40 --no-cpu-baseline --no-profile > $d/log.txt 2>&1
  python tools/pmc_summary.py $d/pmc_results.db 2>&1 | grep "k_fb_slab8\|k_gemm_lds"
done

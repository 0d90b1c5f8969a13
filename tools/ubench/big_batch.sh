timeout 300 python -m pytest tests/test_gpu_update.py -q -x -k "golden or variants or track" 2>&1 | tail -5
for b in 256 1024 4096; do for x in 1 0; do
  v=$(BATCH=$b RLARM_GEMM_XCD=$x python tools/ubench/notorch_cycle.py 2>&1 | grep "n_batches=40")
  echo "batch $b xcd $x: $v"
done; done
for b in 256 1024; do BATCH=$b RLARM_LIB=$PWD/rl_arm_under_sparse_reward_amd/librlarm_tl.so python tools/ubench/notorch_cycle.py 2>&1 | grep "timeline dW\|us/step"; done

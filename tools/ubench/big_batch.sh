for b in 256 128 384 448 512 768; do for pf in 0 1 0 1; do
  v=$(BATCH=$b RLARM_FB_PREFETCH=$pf python tools/ubench/notorch_cycle.py 2>&1 | grep "n_batches=40")
  echo "batch $b prefetch $pf: $v"
done; done

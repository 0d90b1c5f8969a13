for b in 128 384 448 768; do for x in 1 0 1 0; do
  v=$(BATCH=$b RLARM_FB_XCD=$x python tools/ubench/notorch_cycle.py 2>&1 | grep "n_batches=40")
  echo "batch $b fb_xcd $x: $v"
done; done

for rep in 1 2 3; do
  v=$(BATCH=256 python tools/ubench/notorch_cycle.py 2>&1 | grep "n_batches=40")
  echo "batch 256: $v"
done

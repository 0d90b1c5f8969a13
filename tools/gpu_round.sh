#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel trace.  Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_full.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_gpu_full.log > gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
python bench.py --steps 4000 --warmup 400 > gpurun_out/bench.log 2>&1
tail -1 gpurun_out/bench.log > gpurun_out/bench.json
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o trace -- python bench.py --steps 2000 --warmup 400 --no-cpu-baseline --no-profile > gpurun_out/prof_bench.log 2>&1
rm -rf gpurun_out/prof_csv && mkdir -p gpurun_out/prof_csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_csv -o trace -- python bench.py --steps 2000 --warmup 400 --no-cpu-baseline --no-profile > gpurun_out/prof_csv/log.txt 2>&1
ls -R gpurun_out/prof | head -30
cat gpurun_out/pytest_gpu.log gpurun_out/smoke.log
cat gpurun_out/bench.json

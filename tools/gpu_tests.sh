#!/bin/bash
# The whole GPU suite through gpurun (what the driver runs at round end): /usr/local/graft/bin/gpurun --timeout 3000 -- bash tools/gpu_tests.sh
set -u
export TMPDIR=/tmp
O=gpurun_out/tests; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q -rs > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed|SKIPPED|FAILED|ERROR" $O/pytest.log | tail -25

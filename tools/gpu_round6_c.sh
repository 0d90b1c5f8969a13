#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r06c; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q -rs > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed|SKIPPED|FAILED|ERROR" $O/pytest.log | tail -25

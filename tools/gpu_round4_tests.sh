#!/bin/bash
# full GPU suite, log to gpurun_out/r04/tests.log
set -u
mkdir -p gpurun_out/r04; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q -rfEs 2>&1 | tail -40 | tee gpurun_out/r04/tests.log

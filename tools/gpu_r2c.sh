#!/bin/bash
set -u
export TMPDIR=/tmp
python tools/ubench/cycle_drift.py --mode cycle 2>&1 | tail -3
python tools/ubench/cycle_drift.py --mode nogc 2>&1 | tail -3

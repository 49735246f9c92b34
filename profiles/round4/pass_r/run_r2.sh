#!/bin/bash
# which test hangs: verbose, unbuffered, a limit per test
set -u
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
timeout 1200 python -u -m pytest tests -m gpu -x -v --timeout 120 2>&1 | tail -n 60 > $O/pytest_v.txt; tail -30 $O/pytest_v.txt

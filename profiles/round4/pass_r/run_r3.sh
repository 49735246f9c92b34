#!/bin/bash
# the GPU suite four times over, a limit per test: does anything hang?
set -u
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
for k in 1 2 3 4; do
timeout 900 python -u -m pytest tests -m gpu -x -v --timeout 120 2>&1 | tail -n 12 > $O/pytest_v$k.txt; tail -4 $O/pytest_v$k.txt
done

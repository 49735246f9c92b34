#!/bin/bash
# round 4, pass S5: the GPU suite ten times over (a limit per test): anything that hangs or fails once in a while?
set -u
export TMPDIR=/tmp
O=gpurun_out/r4s; mkdir -p $O
for k in 1 2 3 4 5 6 7 8 9 10; do
timeout 900 python -u -m pytest tests -m gpu -x -q --timeout 200 2>&1 | tail -n 2 | tr '\n' ' ' | tee -a $O/soak.txt; echo | tee -a $O/soak.txt
done

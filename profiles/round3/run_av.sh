#!/bin/bash
# round 3, pass AV: sorted passes: how often a block is re-sorted (VX_SORT_PERIOD frames of its stream; 1 = every frame); two against three frames in flight
set -u
export TMPDIR=/tmp
python3 -m pytest tests -m gpu -x -q -k 'kernel_versions or sorted_passes' 2>&1 | grep -E 'passed|failed' | cut -c1-200
for i in 1 2; do for per in 1 4 8 16 64; do for f in csvo esvo; do VX_SORT_PERIOD=$per timeout 300 python3 bench.py --format $f --no-cpu-baseline --no-sd500 --repeats 9 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('period $per $f', d['value'], d['ms_per_step'], d['roofline']['kernel_exclusive_ms'])"; done; done; done

#!/bin/bash
# round 3, pass AS: quick parity + C3 after a change to the sorted builds
set -u
export TMPDIR=/tmp
python3 -m pytest tests -m gpu -x -q -k 'kernel_versions or full_size or sharded or heightfield' 2>&1 | grep -E 'passed|failed' | cut -c1-200
for i in 1 2; do for f in csvo esvo; do timeout 300 python3 bench.py --format $f --no-cpu-baseline --no-sd500 --repeats 9 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'], d['roofline']['kernel_exclusive_ms'], d['roofline']['kernel_exclusive_ms_timed_policy'])"; done; done

#!/bin/bash
# round 4, pass T2 (experiment): the refill's tile arithmetic cut down (columns only, stretches of 16 by shifts): does the smaller kernel show?
set -u
export TMPDIR=/tmp
O=gpurun_out/r4t; mkdir -p $O
for rep in 1 2 3; do for fmt in csvo esvo; do
  timeout 600 python bench.py --format $fmt --no-cpu-baseline --no-extras > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('$fmt: in flight', d['ms_per_step'], d['value'], 'one at a time (HIP bracket)', d['roofline'].get('kernel_exclusive_ms'))" | tee -a $O/lean.txt
done; done

#!/bin/bash
# round 4, pass T4 (experiment): every other frame in flight handed out backwards (a frame starts where the one before it ends)
set -u
export TMPDIR=/tmp
O=gpurun_out/r4t; mkdir -p $O
for rep in 1 2 3; do for alt in 0 1; do for fmt in csvo; do
  VX_ALTERNATE=$alt timeout 600 python bench.py --format $fmt --no-cpu-baseline --no-extras > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('$fmt alternate $alt: in flight', d['ms_per_step'], d['value'])" | tee -a $O/alternate.txt
done; done; done

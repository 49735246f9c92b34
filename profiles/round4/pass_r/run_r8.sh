#!/bin/bash
# round 4, pass R8: stretches that divide the launch evenly among the dispensers (32640 sub-tiles = 8 x 4080), two rounds
set -u
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
for rep in 1 2; do for fmt in csvo; do for s in 16 48 240 510 1020 2040 4080; do
  VX_QUEUE_STRIPE=$s VX_HOT_FIRST=0 timeout 600 python bench.py --format $fmt --no-cpu-baseline --no-extras > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('$fmt stripe $s (no cost order): two in flight', d['ms_per_step'], 'one at a time (HIP bracket)', d['roofline'].get('kernel_exclusive_ms'))" | tee -a $O/stripes3.txt
done; done; done

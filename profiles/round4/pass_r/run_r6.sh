#!/bin/bash
# round 4, pass R6: the length of the stretches (sub-tiles; 1 = round 3's every-eighth, 960 = a row of tiles at 1080p, 0 = an eighth of the launch), three rounds
set -u
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
for rep in 1 2 3; do for fmt in csvo esvo; do for s in 1 16 960 1920 0; do
  VX_QUEUE_STRIPE=$s VX_HOT_FIRST=0 timeout 600 python bench.py --format $fmt --no-cpu-baseline --no-extras > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('$fmt stripe $s (no cost order): two in flight', d['ms_per_step'], 'one at a time (HIP bracket)', d['roofline'].get('kernel_exclusive_ms'))" | tee -a $O/stripes2.txt
done; done; done

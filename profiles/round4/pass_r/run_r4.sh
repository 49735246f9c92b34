#!/bin/bash
# round 4, pass R4: regions against every-eighth-sub-tile, one frame at a time (with and without the cost-ordered table) and two frames in flight
set -u
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
for fmt in csvo esvo; do for reg in 0 1; do for hot in 1 0; do
  VX_QUEUE_REGIONS=$reg VX_HOT_FIRST=$hot timeout 600 python bench.py --format $fmt --no-cpu-baseline --no-extras > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('$fmt regions $reg hot_first $hot: two in flight', d['ms_per_step'], 'one at a time (HIP bracket)', d['roofline'].get('kernel_exclusive_ms'))" | tee -a $O/regions.txt
done; done; done

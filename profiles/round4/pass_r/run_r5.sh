#!/bin/bash
# round 4, pass R5: the length of the stretches the sub-tiles are dealt out in (log2; 0 = round 3's every-eighth), two frames in flight and one at a time without the cost-ordered table
set -u
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
for fmt in csvo esvo; do for s in 0 4 6 8 10 12 -1; do
  VX_QUEUE_STRIPE=$s VX_HOT_FIRST=0 timeout 600 python bench.py --format $fmt --no-cpu-baseline --no-extras > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('$fmt stripe_shift $s (no cost order): two in flight', d['ms_per_step'], 'one at a time (HIP bracket)', d['roofline'].get('kernel_exclusive_ms'))" | tee -a $O/stripes.txt
done; done
VX_HOT_FIRST=1 timeout 600 python bench.py --format csvo --no-cpu-baseline --no-extras > $O/b.json 2>/dev/null
python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('csvo defaults: two in flight', d['ms_per_step'], 'one at a time (HIP bracket)', d['roofline'].get('kernel_exclusive_ms'))" | tee -a $O/stripes.txt

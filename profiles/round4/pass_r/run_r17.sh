#!/bin/bash
# round 4, pass R17: service_min once a wave has found the queue empty (its last sub-tiles out of lockstep), one frame at a time and in flight
set -u
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
for rep in 1 2; do for t in 64 32 16 8 4 1; do
  VX_TAIL_SERVICE_MIN=$t timeout 600 python bench.py --format csvo --no-cpu-baseline --no-extras > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('csvo tail_service_min $t: one at a time (HIP bracket)', d['roofline'].get('kernel_exclusive_ms'), 'in flight', d['ms_per_step'])" | tee -a $O/tail.txt
done; done

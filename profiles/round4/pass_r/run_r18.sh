#!/bin/bash
# round 4, pass R18: service_min / refill_min again, under the new tile order
set -u
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
for combo in "64 4" "64 1" "64 16" "64 32" "48 4" "32 4" "56 4" "60 4"; do
  set -- $combo
  VX_SERVICE_MIN=$1 VX_REFILL_MIN=$2 timeout 600 python bench.py --format csvo --no-cpu-baseline --no-extras > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('csvo service_min $1 refill_min $2: in flight', d['ms_per_step'], 'one at a time (HIP bracket)', d['roofline'].get('kernel_exclusive_ms'))" | tee -a $O/service.txt
done

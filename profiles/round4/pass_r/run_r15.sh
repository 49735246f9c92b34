#!/bin/bash
# round 4, pass R15: one frame at a time -- which sub-tiles the cost order puts first (classes of `step` iterations; the classes below `first` as one, in queue order)
set -u
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
for rep in 1 2; do for combo in "16 0" "16 4" "16 6" "16 8" "16 10" "16 12" "32 0" "32 3" "32 5" "64 2"; do
  set -- $combo
  VX_ORDER_STEP=$1 VX_ORDER_FIRST_CLASS=$2 timeout 600 python bench.py --format csvo --no-cpu-baseline --no-extras > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('csvo order step $1 first class $2: one at a time (HIP bracket)', d['roofline'].get('kernel_exclusive_ms'), 'in flight', d['ms_per_step'])" | tee -a $O/order.txt
done; done
VX_HOT_FIRST=0 timeout 600 python bench.py --format csvo --no-cpu-baseline --no-extras > $O/b.json 2>/dev/null
python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('csvo no cost order: one at a time (HIP bracket)', d['roofline'].get('kernel_exclusive_ms'))" | tee -a $O/order.txt

#!/bin/bash
# round 4, pass R9: tiles numbered down the columns (a stretch of the queue = a vertical band: sky and ground in every XCD's share), two rounds
set -u
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
for rep in 1 2; do for fmt in csvo; do for col in 0 1; do for s in 16 1020 4080; do
  VX_QUEUE_COLUMNS=$col VX_QUEUE_STRIPE=$s VX_HOT_FIRST=0 timeout 600 python bench.py --format $fmt --no-cpu-baseline --no-extras > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('$fmt columns $col stripe $s (no cost order): two in flight', d['ms_per_step'], 'one at a time (HIP bracket)', d['roofline'].get('kernel_exclusive_ms'))" | tee -a $O/columns.txt
done; done; done; done
VX_QUEUE_COLUMNS=1 timeout 600 python -u -m pytest tests/test_hip_parity.py -m gpu -x -q --timeout 300 -k "moving or sizes or edges or cost_ordered" 2>&1 | tail -3 | tee -a $O/columns.txt

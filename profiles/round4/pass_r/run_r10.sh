#!/bin/bash
# round 4, pass R10: how the queue's tile numbers lie on the screen -- 0 along the rows, 1 down the columns, 2 a stride of 0.618 of the tiles -- two rounds
set -u
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
VX_TILE_NUMBERING=2 timeout 900 python -u -m pytest tests/test_hip_parity.py tests/test_baseline_configs.py -m gpu -x -q --timeout 300 -k "moving or sizes or edges or cost_ordered or c3 or C3 or versions" 2>&1 | tail -3 | tee -a $O/numbering.txt
for rep in 1 2; do for fmt in csvo esvo; do for num in 0 1 2; do for s in 16 0; do for hot in 0 1; do
  VX_TILE_NUMBERING=$num VX_QUEUE_STRIPE=$s VX_HOT_FIRST=$hot timeout 600 python bench.py --format $fmt --no-cpu-baseline --no-extras > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('$fmt numbering $num stripe $s hot_first $hot: two in flight', d['ms_per_step'], 'one at a time (HIP bracket)', d['roofline'].get('kernel_exclusive_ms'))" | tee -a $O/numbering.txt
done; done; done; done; done

#!/bin/bash
# round 4, pass R12: tile numbering 1 with strips of W columns (1 = down the columns), stretches of a tile (16) or an eighth of the launch (0)
set -u
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
VX_TILE_STRIP=4 timeout 900 python -u -m pytest tests/test_hip_parity.py tests/test_baseline_configs.py -m gpu -x -q --timeout 300 -k "moving or sizes or edges or cost_ordered or c3 or C3 or versions or sharded" 2>&1 | tail -2 | tee -a $O/strips.txt
for w in 1 2 4 8; do for s in 16 0; do
  VX_TILE_STRIP=$w VX_QUEUE_STRIPE=$s timeout 600 python bench.py --format csvo --no-cpu-baseline --no-extras > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('csvo strip $w stripe $s: C3 two in flight', d['ms_per_step'], 'one at a time (cost order; HIP bracket)', d['roofline'].get('kernel_exclusive_ms'))" | tee -a $O/strips.txt
  VX_TILE_STRIP=$w VX_QUEUE_STRIPE=$s timeout 900 python profiles/configs_bench.py --format csvo --configs C2 C4-d13 C4 C5 2>/dev/null | grep -h '"config"' | python -c "
import sys, json
print('csvo strip $w stripe $s:', ' '.join('%s %s' % (json.loads(l)['config'], json.loads(l)['ms_per_frame']) for l in sys.stdin))
" | tee -a $O/strips.txt
done; done

#!/bin/bash
# round 4, pass T: primary rays that miss the box around the world's chunks are not traversed (VX_BOX_CULL=0: they are) -- bench and configurations, both formats
set -u
export TMPDIR=/tmp
O=gpurun_out/r4t; mkdir -p $O; rm -f $O/*
for rep in 1 2; do for cull in 1 0; do for fmt in csvo esvo; do
  VX_BOX_CULL=$cull timeout 600 python bench.py --format $fmt --no-cpu-baseline --no-extras > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('$fmt box cull $cull: in flight', d['ms_per_step'], d['value'], 'one at a time (HIP bracket)', d['roofline'].get('kernel_exclusive_ms'))" | tee -a $O/cull.txt
done; done; done
for cull in 1 0; do for fmt in csvo esvo; do
  VX_BOX_CULL=$cull timeout 900 python profiles/configs_bench.py --format $fmt 2>/dev/null | grep -h '"config"' | python -c "
import sys, json
print('$fmt box cull $cull:', ' '.join('%s %s' % (json.loads(l)['config'], json.loads(l)['ms_per_frame']) for l in sys.stdin))
" | tee -a $O/cull.txt
done; done

#!/bin/bash
# round 4, pass T3 (experiment): the wave's issue priority (s_setprio) -- 3 in the traversal loop and 0 in the service phases, or the other way round, against none
set -u
export TMPDIR=/tmp
O=gpurun_out/r4t; mkdir -p $O
for rep in 1 2; do for lib in lib lib/lib_prio3 lib/lib_prio0; do for fmt in csvo esvo; do
  VX_LIB_DIR=$PWD/voxel-rs_amd/$lib timeout 600 python bench.py --format $fmt --no-cpu-baseline --no-extras > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('$lib $fmt: in flight', d['ms_per_step'], d['value'], 'one at a time (HIP bracket)', d['roofline'].get('kernel_exclusive_ms'))" | tee -a $O/prio.txt
done; done; done

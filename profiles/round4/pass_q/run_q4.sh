#!/bin/bash
# round 4, pass Q4: frames in flight 1 vs 2 on the 4K depth-13 terrain, both formats (does a kernel with scratch overlap its successor as well as one without?)
set -u
export TMPDIR=/tmp
O=gpurun_out/r4q; mkdir -p $O
for fmt in esvo csvo; do for fif in 1 2 3; do
  VX_FRAMES_IN_FLIGHT=$fif timeout 600 python profiles/configs_bench.py --format $fmt --configs C4-d13 2>/dev/null | grep -h '"config"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$fmt fif $fif', d['config'], d['ms_per_frame'])
" | tee -a $O/fif.txt
done; done

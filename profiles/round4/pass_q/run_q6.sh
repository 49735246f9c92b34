#!/bin/bash
# round 4, pass Q6: one frame at a time, screen order, production library against the timeline build with its instrumentation idle (4K depth 14 and 13, both formats)
set -u
export TMPDIR=/tmp
O=gpurun_out/r4q; mkdir -p $O
export VX_FRAMES_IN_FLIGHT=1 VX_HOT_FIRST=0
for lib in lib lib/lib_tl; do for fmt in esvo csvo; do
  VX_LIB_DIR=$PWD/voxel-rs_amd/$lib timeout 600 python profiles/configs_bench.py --format $fmt --configs C4-d13 C4 2>/dev/null | grep -h '"config"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib $fmt fif1 hot0', d['config'], d['ms_per_frame'])
" | tee -a $O/tl_vs_prod.txt
done; done

#!/bin/bash
# round 4, pass Q5 (experiment): walks cut to one / two iterations and dropped (timing only): what the walk phases cost beside their iterations
set -u
export TMPDIR=/tmp
O=gpurun_out/r4q; mkdir -p $O
for combo in "1 1" "2 1"; do
  set -- $combo
  VX_WALK_CAP=$1 VX_WALK_DROP=$2 timeout 600 python profiles/configs_bench.py --format csvo --configs C4-d13 C4 2>/dev/null | grep -h '"config"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('cap $1 drop $2', d['config'], d['ms_per_frame'], d.get('of_which_started_over'), d.get('iterations_on_bytes_per_frame'))
" | tee -a $O/cap.txt
done
timeout 600 python profiles/configs_bench.py --format esvo --configs C4-d13 C4 2>/dev/null | grep -h '"config"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('esvo', d['config'], d['ms_per_frame'])
" | tee -a $O/cap.txt

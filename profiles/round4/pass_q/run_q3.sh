#!/bin/bash
# round 4, pass Q3 (experiment): what a cap on a walk phase's iterations would save -- capped shadow rays DROPPED (wrong pixels: timing only) or listed for the world's bytes
set -u
export TMPDIR=/tmp
O=gpurun_out/r4q; mkdir -p $O
for combo in "0 0" "3 1" "4 1" "5 1" "6 1" "8 1" "12 1" "4 0" "6 0"; do
  set -- $combo
  VX_WALK_CAP=$1 VX_WALK_DROP=$2 timeout 600 python profiles/configs_bench.py --format csvo --configs C4-d13 C4 2>/dev/null | grep -h '"config"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('cap $1 drop $2', d['config'], d['ms_per_frame'], d.get('of_which_started_over'), d.get('iterations_on_bytes_per_frame'))
" | tee -a $O/cap.txt
done

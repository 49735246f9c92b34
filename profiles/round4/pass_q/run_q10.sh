#!/bin/bash
# round 4, pass Q10: a cap on a shadow ray's walk, the capped rays listed for the world's bytes (what the walk gives up on anyway): the cap swept, twice
set -u
export TMPDIR=/tmp
O=gpurun_out/r4q; mkdir -p $O
for rep in 1 2; do for cap in 0 5 6 7 8 10; do
  VX_WALK_CAP=$cap timeout 600 python profiles/configs_bench.py --format csvo --configs C4-d13 C4 C5 2>/dev/null | grep -h '"config"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('cap $cap (bytes):', d['config'], d['ms_per_frame'], d.get('of_which_started_over'), d.get('iterations_on_bytes_per_frame'))
" | tee -a $O/cap_bytes.txt
done; done

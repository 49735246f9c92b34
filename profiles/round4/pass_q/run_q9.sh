#!/bin/bash
# round 4, pass Q9: as Q8, the rerun's lanes walking together
set -u
export TMPDIR=/tmp
O=gpurun_out/r4q; mkdir -p $O
VX_WALK_CAP=4 timeout 1500 python -m pytest tests/test_baseline_c4_c5.py tests/test_hip_parity.py -m gpu -x -q -k "deep or c4 or c5 or inside or versions" > $O/pytest_cap4b.txt 2>&1; echo "pytest rc $?" >> $O/pytest_cap4b.txt; tail -3 $O/pytest_cap4b.txt
for cap in 0 4 6; do
  VX_WALK_CAP=$cap timeout 600 python profiles/configs_bench.py --format csvo --configs C4-d13 C4 2>/dev/null | grep -h '"config"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('cap $cap, capped rays on the image again, walking together, fast step:', d['config'], d['ms_per_frame'], d.get('of_which_started_over'), d.get('iterations_on_bytes_per_frame'))
" | tee -a $O/cap_image.txt
done

#!/bin/bash
# round 4, pass Q8: capped walks, the capped shadow rays run on the image again at the end of the wave's life (walk at full length): parity, then the cap swept
set -u
export TMPDIR=/tmp
O=gpurun_out/r4q; mkdir -p $O
VX_WALK_CAP=4 timeout 1500 python -m pytest tests/test_baseline_c4_c5.py tests/test_hip_parity.py -m gpu -x -q -k "deep or c4 or c5 or inside or versions" > $O/pytest_cap4.txt 2>&1; echo "pytest rc $?" >> $O/pytest_cap4.txt; tail -3 $O/pytest_cap4.txt
VX_WALK_CAP=2 timeout 1500 python -m pytest tests/test_baseline_c4_c5.py tests/test_hip_parity.py -m gpu -x -q -k "deep or c4 or c5 or inside" > $O/pytest_cap2.txt 2>&1; echo "pytest rc $?" >> $O/pytest_cap2.txt; tail -3 $O/pytest_cap2.txt
for cap in 0 3 4 5 6 8; do
  VX_WALK_CAP=$cap timeout 600 python profiles/configs_bench.py --format csvo --configs C4-d13 C4 2>/dev/null | grep -h '"config"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('cap $cap, capped rays on the image again:', d['config'], d['ms_per_frame'], d.get('of_which_started_over'), d.get('iterations_on_bytes_per_frame'))
" | tee -a $O/cap_image.txt
done

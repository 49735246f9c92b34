#!/bin/bash
# round 4, pass Q: the walk inside a voxel as one stretch of code per iteration (selects instead of the if / else tree)
set -u
export TMPDIR=/tmp
O=gpurun_out/r4q; mkdir -p $O; rm -f $O/*
timeout 1500 python -m pytest tests/test_baseline_c4_c5.py tests/test_hip_parity.py -m gpu -x -q -k "deep or c4 or c5 or inside or versions" > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -3 $O/pytest.txt
timeout 900 python profiles/configs_bench.py --format csvo --configs C4-d13 C4 C5 > $O/configs_csvo.json 2> $O/configs_csvo.err
grep -h '"config"' $O/configs_csvo.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], d['ms_per_frame'], d.get('rays_led_into_a_voxel_per_frame'), d.get('of_which_started_over'), d.get('excursion_phases_per_frame'), d.get('iterations_on_bytes_per_frame'))
" | tee $O/summary.txt

#!/bin/bash
# round 4, pass D: rays the walk gives up on are listed and run on the bytes (image-only renders); a cap on the stragglers of a walk phase
set -u
export TMPDIR=/tmp
O=gpurun_out/r4d; mkdir -p $O; rm -f $O/*
export VX_FOREIGN_MIN=1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -3 $O/pytest.txt
# the list as the common path: every walk longer than two iterations given up
VX_WALK_CAP=2 VX_WALK_LANES=64 timeout 900 python -m pytest tests/test_baseline_c4_c5.py tests/test_hip_parity.py -m gpu -x -q -k "deep or c4 or c5 or inside or versions" > $O/pytest_cap2.txt 2>&1; echo "pytest rc $?" >> $O/pytest_cap2.txt; tail -3 $O/pytest_cap2.txt
for combo in "4294967295 0" "3 2" "4 4" "4 8" "6 4" "6 8" "8 2" "5 16"; do
  set -- $combo
  VX_WALK_CAP=$1 VX_WALK_LANES=$2 timeout 600 python profiles/configs_bench.py --format csvo --configs C4-d13 C4 > $O/configs_cap$1_lanes$2.json 2> $O/configs_cap$1_lanes$2.err
  grep -h '"config"' $O/configs_cap$1_lanes$2.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('cap $1 lanes $2', d['config'], d['ms_per_frame'], d['rays_led_into_a_voxel_per_frame'], d['of_which_started_over'], d['excursion_phases_per_frame'], d['iterations_on_bytes_per_frame'])
" | tee -a $O/summary.txt
done

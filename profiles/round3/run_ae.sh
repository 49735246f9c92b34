#!/bin/bash
# round 3, pass AE: deep CSVO worlds: phantom leaves of opaque blocks inside the walk are hits without their sample (as in the kernel's own leaf tests)
set -u
export TMPDIR=/tmp
O=gpurun_out/r3ae; mkdir -p $O; rm -f $O/*
timeout 900 python3 -m pytest tests -m gpu -x -q -k "deep_world or inside or c4 or c5 or kernel_versions or streamed or full_size or heightfield_frame" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed|rc=" $O/pytest.log | cut -c1-200
for c in C4-d13 C4 C5; do timeout 600 python3 profiles/configs_bench.py --format csvo --configs $c 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], d['ms_per_frame'], 'phases', d['excursion_phases_per_frame'], 'given up', d['of_which_started_over'])"; done | tee $O/opaque_walk.txt
VX_TIMELINE=1 timeout 300 python3 profiles/timeline.py --format csvo --depth 13 --width 3840 --height 2160 --hot 0 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('service us', d['us_in_service_phases_per_wave'][2], 'phases', d['service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'cycles/trip', d['cycles_per_trip_mean'], 'trips', d['loop_trips_per_wave'][2])" | tee -a $O/opaque_walk.txt
VX_TIMELINE=1 VX_TIMELINE_PART=5 timeout 300 python3 profiles/timeline.py --format csvo --depth 13 --width 3840 --height 2160 --hot 0 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('walk part us', d['us_in_service_phases_per_wave'][2])" | tee -a $O/opaque_walk.txt

#!/bin/bash
# round 3, pass X: deep CSVO worlds: shadow rays that end inside their voxel are held until the next phase that walks nobody (VX_HOLD_RESOLVED)
set -u
export TMPDIR=/tmp
O=gpurun_out/r3x; mkdir -p $O; rm -f $O/*
timeout 900 python3 -m pytest tests -m gpu -x -q -k "kernel_versions or deep_world or inside or c4 or c5 or streamed" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 3 $O/pytest.log | cut -c1-200
for h in 0 1; do for c in C4-d13 C4; do VX_HOLD_RESOLVED=$h timeout 600 python3 profiles/configs_bench.py --format csvo --configs $c 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('hold', $h, d['config'], d['ms_per_frame'], 'phases', d['excursion_phases_per_frame'], 'given up', d['of_which_started_over'])"; done; done | tee $O/hold.txt
for h in 0 1; do VX_HOLD_RESOLVED=$h VX_TIMELINE=1 timeout 300 python3 profiles/timeline.py --format csvo --depth 13 --width 3840 --height 2160 --hot 0 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('hold', $h, 'service us', d['us_in_service_phases_per_wave'][2], 'phases', d['service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'cycles/trip', d['cycles_per_trip_mean'], 'trips', d['loop_trips_per_wave'][2])"; done | tee -a $O/hold.txt

#!/bin/bash
# round 3, pass AA: deep CSVO worlds at three waves per SIMD (VX_DEEP_WAVES=3: 168 registers, 2 spilled instead of 52-61; 12 waves per CU)
set -u
export TMPDIR=/tmp
O=gpurun_out/r3aa; mkdir -p $O; rm -f $O/*
VX_DEEP_WAVES=3 timeout 900 python3 -m pytest tests -m gpu -x -q -k "deep_world or inside or c4 or c5" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed|rc=" $O/pytest.log | cut -c1-200
for w in 4 3; do for c in C4-d13 C4 C5; do VX_DEEP_WAVES=$w timeout 600 python3 profiles/configs_bench.py --format csvo --configs $c 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('deep_waves', $w, d['config'], d['ms_per_frame'], 'phases', d['excursion_phases_per_frame'], 'given up', d['of_which_started_over'])"; done; done | tee $O/deep_waves.txt
for w in 4 3; do VX_DEEP_WAVES=$w VX_TIMELINE=1 timeout 300 python3 profiles/timeline.py --format csvo --depth 13 --width 3840 --height 2160 --hot 0 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('deep_waves', $w, 'service us', d['us_in_service_phases_per_wave'][2], 'phases', d['service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'cycles/trip', d['cycles_per_trip_mean'], 'trips', d['loop_trips_per_wave'][2])"; done | tee -a $O/deep_waves.txt

#!/bin/bash
# round 3, pass AD: deep CSVO worlds: what a lane carries through the walk inside a voxel put away by hand around it (VX_PARK_AROUND_WALK=1, voxel-rs_amd/lib)
# against the build without (voxel-rs_amd/lib_np); the code was removed after the measurement
set -u
export TMPDIR=/tmp
O=gpurun_out/r3ad; mkdir -p $O; rm -f $O/*
timeout 900 python3 -m pytest tests -m gpu -x -q -k "deep_world or inside or c4 or c5 or kernel_versions or streamed" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed|rc=" $O/pytest.log | cut -c1-200
for L in lib_np lib lib_np lib; do for c in C4-d13 C4; do VX_LIB_DIR=voxel-rs_amd/$L timeout 600 python3 profiles/configs_bench.py --format csvo --configs $c 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['config'], d['ms_per_frame'], 'phases', d['excursion_phases_per_frame'], 'given up', d['of_which_started_over'])"; done; done | tee $O/park.txt
for L in lib_np lib; do VX_LIB_DIR=voxel-rs_amd/$L VX_TIMELINE=1 timeout 300 python3 profiles/timeline.py --format csvo --depth 13 --width 3840 --height 2160 --hot 0 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', 'service us', d['us_in_service_phases_per_wave'][2], 'phases', d['service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'cycles/trip', d['cycles_per_trip_mean'], 'trips', d['loop_trips_per_wave'][2])"; done | tee -a $O/park.txt

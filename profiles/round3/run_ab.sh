#!/bin/bash
# round 3, pass AB: deep CSVO worlds, the walk inside a voxel as a real call (a build with that code, since removed, in voxel-rs_amd/lib_wc) against the
# inlined walk (voxel-rs_amd/lib); and the cost notes by wave (one atomic per sub-tile and phase) on C3, one frame at a time
set -u
export TMPDIR=/tmp
O=gpurun_out/r3ab; mkdir -p $O; rm -f $O/*
VX_LIB_DIR=voxel-rs_amd/lib_wc timeout 900 python3 -m pytest tests -m gpu -x -q -k "deep_world or inside or c4 or c5 or kernel_versions" > $O/pytest_wc.log 2>&1; echo "rc=$?" >> $O/pytest_wc.log; grep -E "passed|failed|rc=" $O/pytest_wc.log | cut -c1-200
for L in lib lib_wc; do for c in C4-d13 C4 C5; do VX_LIB_DIR=voxel-rs_amd/$L timeout 600 python3 profiles/configs_bench.py --format csvo --configs $c 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['config'], d['ms_per_frame'], 'phases', d['excursion_phases_per_frame'], 'given up', d['of_which_started_over'])"; done; done | tee $O/walk_call.txt
for L in lib lib_wc; do VX_LIB_DIR=voxel-rs_amd/$L VX_TIMELINE=1 timeout 300 python3 profiles/timeline.py --format csvo --depth 13 --width 3840 --height 2160 --hot 0 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', 'service us', d['us_in_service_phases_per_wave'][2], 'phases', d['service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'cycles/trip', d['cycles_per_trip_mean'], 'trips', d['loop_trips_per_wave'][2])"; done | tee -a $O/walk_call.txt
timeout 600 python3 -m pytest tests -m gpu -x -q -k "cost_ordered or kernel_versions or full_size" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed|rc=" $O/pytest.log | cut -c1-200
for f in csvo esvo; do timeout 300 python3 bench.py --format $f --no-cpu-baseline --no-sd500 --repeats 9 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'], d['roofline']['kernel_exclusive_ms'], d['roofline']['kernel_exclusive_ms_timed_policy'])"; done | tee $O/notes_by_wave.txt

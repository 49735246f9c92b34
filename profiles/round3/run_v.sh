#!/bin/bash
# round 3, pass V: the shape of a sub-tile (64 pixels, one per lane): 8x8 (Morton), 16x4, 32x2 -- three builds of the library on one box
set -u
export TMPDIR=/tmp
O=gpurun_out/r3v; mkdir -p $O; rm -f $O/*
for L in lib_w16 lib_w32; do VX_LIB_DIR=voxel-rs_amd/$L timeout 600 python3 -m pytest tests -m gpu -x -q -k "kernel_versions or heightfield_frame or full_size or edges or sharded or cost_ordered" 2>&1 | tail -n 2 | cut -c1-200; done
for i in 1 2 3; do for f in csvo esvo; do
  for L in lib lib_w16 lib_w32; do VX_LIB_DIR=voxel-rs_amd/$L timeout 300 python3 bench.py --format $f --no-cpu-baseline --no-sd500 --repeats 9 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L $f', d['value'], d['ms_per_step'], d['roofline']['kernel_exclusive_ms'], d['roofline']['kernel_exclusive_ms_timed_policy'])"; done
done; done | tee $O/subtile_shape.txt
for L in lib lib_w16 lib_w32; do VX_LIB_DIR=voxel-rs_amd/$L VX_TIMELINE=1 timeout 200 python3 profiles/timeline.py --format csvo --hot 0 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', 'trips', d['loop_trips_per_wave'][2], 'cycles/trip', d['cycles_per_trip_mean'], 'service us', d['us_in_service_phases_per_wave'][2], 'phases', d['service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'kernel', d['kernel_us'])"; done | tee -a $O/subtile_shape.txt

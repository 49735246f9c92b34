#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r3p; mkdir -p $O; rm -f $O/*
for f in csvo; do for part in 0 1 2 3 4; do
  VX_TIMELINE_PART=$part VX_TIMELINE=1 timeout 200 python3 profiles/timeline.py --format $f --hot 0 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f hot 0 part', $part, 'us per wave p10/p50/p90:', d['us_in_service_phases_per_wave'][1:4], 'phases', d['service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'kernel', d['kernel_us'], 'cycles/trip', d['cycles_per_trip_mean'], 'trips', d['loop_trips_per_wave'][2])" | tee -a $O/parts_hot0.txt
done; done

#!/bin/bash
# round 3, pass L: service-phase parts at HEAD (service_min 56, opaque set), frames in flight 2/3/4, waves per CU
set -u
O=gpurun_out/r3l; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
for f in csvo; do for part in 0 1 2 3 4; do
  VX_TIMELINE_PART=$part VX_TIMELINE=1 timeout 200 python3 profiles/timeline.py --format $f 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f part', $part, 'us per wave p10/p50/p90:', d['us_in_service_phases_per_wave'][1:4], 'phases', d['service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'kernel', d['kernel_us'], 'cycles/trip', d['cycles_per_trip_mean'], 'trips', d['loop_trips_per_wave'][2])" >> $O/parts.txt
done; done
cat $O/parts.txt
timeout 600 python3 profiles/sweep.py --format csvo --rounds 5 --steps 20 --configs "f=2" "f=3" "f=4" "f=2,w=12" "f=2,w=14" "f=2,s=60" "f=2,s=64" "f=3,s=60" 2>&1 | grep -v "^counters" | tee $O/sweep_csvo.txt

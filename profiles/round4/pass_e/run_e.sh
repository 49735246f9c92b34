#!/bin/bash
# round 4, pass E: a wave's life part by part, deep CSVO (lean walk, foreign_min 1, listed rays) against ESVO, depth 13 and 14
set -u
export TMPDIR=/tmp
O=gpurun_out/r4e; mkdir -p $O; rm -f $O/*
export VX_FOREIGN_MIN=1
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'us per wave p10/p50/p90:', d['us_in_service_phases_per_wave'][1:4], 'phases', d['service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'kernel', d['kernel_us'], 'cycles/trip', d['cycles_per_trip_mean'], 'trips', d['loop_trips_per_wave'][2], 'loop share', d['loop_share_of_wave_life'][2], 'tail', d['tail_us_per_wave'][2:5])"; }
for depth in 13 14; do for f in csvo esvo; do for part in 0 1 2 3 4 5; do
    VX_TIMELINE_PART=$part VX_TIMELINE=1 timeout 300 python3 profiles/timeline.py --format $f --depth $depth --width 3840 --height 2160 --hot 0 2>/dev/null | tail -n 1 | line "$f d$depth part $part" | tee -a $O/parts.txt
done; done; done

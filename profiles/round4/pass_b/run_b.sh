#!/bin/bash
# round 4, pass B: where a wave's life goes in deep CSVO worlds with the lean walk (timeline parts), depth 13 and 14
set -u
export TMPDIR=/tmp
O=gpurun_out/r4b; mkdir -p $O; rm -f $O/*
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'us per wave p10/p50/p90:', d['us_in_service_phases_per_wave'][1:4], 'phases', d['service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'kernel', d['kernel_us'], 'cycles/trip', d['cycles_per_trip_mean'], 'trips', d['loop_trips_per_wave'][2], 'loop share', d['loop_share_of_wave_life'][2], 'tail', d['tail_us_per_wave'][2:5])"; }
for depth in 13 14; do
  for part in 0 1 2 3 4 5; do
    VX_TIMELINE_PART=$part VX_TIMELINE=1 timeout 300 python3 profiles/timeline.py --format csvo --depth $depth --width 3840 --height 2160 --hot 0 2>/dev/null | tail -n 1 | line "csvo d$depth fm40 part $part" | tee -a $O/parts.txt
  done
  for part in 0 5; do
    VX_FOREIGN_MIN=1 VX_TIMELINE_PART=$part VX_TIMELINE=1 timeout 300 python3 profiles/timeline.py --format csvo --depth $depth --width 3840 --height 2160 --hot 0 2>/dev/null | tail -n 1 | line "csvo d$depth fm1 part $part" | tee -a $O/parts.txt
  done
done
for part in 0 2 3; do
  VX_TIMELINE_PART=$part VX_TIMELINE=1 timeout 300 python3 profiles/timeline.py --format esvo --depth 14 --width 3840 --height 2160 --hot 0 2>/dev/null | tail -n 1 | line "esvo d14 part $part" | tee -a $O/parts.txt
done

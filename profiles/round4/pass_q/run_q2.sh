#!/bin/bash
# round 4, pass Q2: what a walk phase's time is made of -- the timeline's part 5 (walks) at 16 and at 8 waves per CU (issue shared by half as many waves)
set -u
export TMPDIR=/tmp
O=gpurun_out/r4q; mkdir -p $O
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'us per wave p10/p50/p90:', d['us_in_service_phases_per_wave'][1:4], 'phases', d['service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'kernel', d['kernel_us'], 'cycles/trip', d['cycles_per_trip_mean'], 'trips', d['loop_trips_per_wave'][2], 'loop share', d['loop_share_of_wave_life'][2], 'tail', d['tail_us_per_wave'][2:5])"; }
for w in 16 8; do
for part in 0 5; do
    VX_WAVES_PER_CU=$w VX_TIMELINE_PART=$part VX_TIMELINE=1 timeout 300 python3 profiles/timeline.py --format csvo --depth 14 --width 3840 --height 2160 --hot 0 2>/dev/null | tail -n 1 | line "csvo d14 waves $w part $part" | tee -a $O/parts.txt
done
done
for w in 16 8; do
    VX_WAVES_PER_CU=$w VX_TIMELINE_PART=0 VX_TIMELINE=1 timeout 300 python3 profiles/timeline.py --format esvo --depth 14 --width 3840 --height 2160 --hot 0 2>/dev/null | tail -n 1 | line "esvo d14 waves $w part 0" | tee -a $O/parts.txt
done

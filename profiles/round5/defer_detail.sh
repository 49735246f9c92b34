#!/bin/bash
# (drove code that was built, measured and reverted: commit 7f12221 -- check it out to run this)
# shadow rays deferred against lockstep, wave by wave (measurement build, C3 ESVO, one frame at a time): trips, phases, where the time is
for cfg in "0 192 32" "1 256 32" "1 256 16" "1 128 32"; do
  set -- $cfg
  for part in 0 1 2 3 4; do
  VX_TIMELINE=1 VX_TIMELINE_PART=$part VX_DEFER_SHADOWS=$1 VX_DEFER_SWITCH=$2 VX_DEFER_SERVICE=$3 python profiles/timeline.py --format esvo 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('defer $1 switch $2 service $3 part $part: kernel_us', d['kernel_us'], 'wave life', d['mean_wave_lifetime_us'], 'trips', d['loop_trips_per_wave'][2], 'cycles/trip', d['cycles_per_trip_mean'], 'loop share', d['loop_share_of_wave_life'][2], 'phases', d['service_phases_per_wave'][2], 'us in part', d['us_in_service_phases_per_wave'][2], 'subtiles', d['subtiles_taken'][2], 'queue empty', d['queue_empty_us'][2], 'exit', d['exit_us'][2], d['exit_us'][5], 'tails', d.get('trips_by_tail_advance_only_push_only_merged'))"
  done
done

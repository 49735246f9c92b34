#!/bin/bash
# lockstep against service_min 56, wave by wave (measurement build, C3 CSVO, one frame at a time): where the quarter goes
for cfg in "64 64" "56 64" "56 4"; do
  set -- $cfg
  for part in 0 1 2 3 4; do
  VX_TIMELINE=1 VX_TIMELINE_PART=$part VX_SERVICE_MIN=$1 VX_REFILL_MIN=$2 python profiles/timeline.py --format csvo 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('service_min $1 refill_min $2 part $part: kernel_us', d['kernel_us'], 'wave life', d['mean_wave_lifetime_us'], 'trips', d['loop_trips_per_wave'][2], 'cycles/trip', d['cycles_per_trip_mean'], 'loop share', d['loop_share_of_wave_life'][2], 'phases', d['service_phases_per_wave'][2], 'us in part', d['us_in_service_phases_per_wave'][2], 'subtiles', d['subtiles_taken'][2], 'queue empty', d['queue_empty_us'][2], 'exit', d['exit_us'][2], d['exit_us'][5])"
  done
done

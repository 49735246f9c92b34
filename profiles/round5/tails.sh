#!/bin/bash
# Which tail of the hand-scheduled loop the trips of a C3 frame take (measurement build, one frame at a time), in lockstep a sub-tile at a time and under
# the policies that fill lanes earlier
for fmt in csvo esvo; do
for cfg in "64 64" "64 4" "56 4" "32 4"; do
  set -- $cfg
  VX_TIMELINE=1 VX_SERVICE_MIN=$1 VX_REFILL_MIN=$2 python profiles/timeline.py --format $fmt 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$fmt service_min $1 refill_min $2: kernel_us', d['kernel_us'], 'trips/wave median', d['loop_trips_per_wave'][2], 'tails adv/push/merged', d['trips_by_tail_advance_only_push_only_merged'], 'cycles/trip', d['cycles_per_trip_mean'], 'phases/wave', d['service_phases_per_wave'][2])"
done; done

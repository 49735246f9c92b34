#!/bin/bash
# round 4, pass R16: the wave timeline of one C3 frame at a time (timeline build), cost order on and off
set -u
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
for hot in 1 0; do for fmt in csvo esvo; do
VX_TIMELINE=1 timeout 300 python3 profiles/timeline.py --format $fmt --hot $hot 2>/dev/null | tail -n 1 > $O/timeline_${fmt}_hot$hot.json
python3 -c "
import json; d=json.load(open('$O/timeline_${fmt}_hot$hot.json'))
keys=['kernel_us','start_us','queue_empty_us','exit_us','tail_us_per_wave','subtiles_taken','service_phases_per_wave','us_in_service_phases_per_wave','mean_wave_lifetime_us','loop_trips_per_wave','cycles_per_trip_mean','loop_share_of_wave_life']
print('$fmt hot $hot', {k:d.get(k) for k in keys})" | tee -a $O/timeline.txt
done; done

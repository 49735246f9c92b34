#!/bin/bash
# round 4, pass R19: the service phases of a C3 frame by part (timeline build; one frame at a time, cost order)
set -u
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
for part in 0 1 2 3 4 5; do
VX_TIMELINE_PART=$part VX_TIMELINE=1 timeout 300 python3 profiles/timeline.py --format csvo --hot 1 2>/dev/null | tail -n 1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('csvo C3 part $part: us per wave p10/p50/p90', d['us_in_service_phases_per_wave'][1:4], 'phases', d['service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'kernel', d['kernel_us'])" | tee -a $O/parts_c3.txt
done

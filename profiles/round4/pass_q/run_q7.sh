#!/bin/bash
# round 4, pass Q7: the walk phases in numbers (timeline build): phases, trips of the walk's loop, cycles -- and the fit cycles = a * phases + b * trips
set -u
export TMPDIR=/tmp
O=gpurun_out/r4q; mkdir -p $O
for w in 16 8; do
VX_WAVES_PER_CU=$w VX_TIMELINE_PART=5 VX_TIMELINE=1 timeout 300 python3 profiles/timeline.py --format csvo --depth 14 --width 3840 --height 2160 --hot 0 2>/dev/null | tail -n 1 > $O/walk_probe_$w.json
python3 -c "
import json; d=json.load(open('$O/walk_probe_$w.json'))
print('waves/CU $w', 'kernel_us', d['kernel_us'], 'walk phases', d['walk_phases_per_wave'][1:4], 'trips', d['walk_trips_per_wave'][1:4], 'cycles', d['walk_cycles_per_wave'][1:4], 'fit [cycles/phase, cycles/trip]', d['walk_fit_cycles_per_phase_and_per_trip'], 'clock', d['clock_mhz'][2])
" | tee -a $O/walk_probe.txt
done

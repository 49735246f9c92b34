#!/bin/bash
# round 3, pass AO: SORTED builds: what a wave's refill part (3) costs with pieces of the machinery switched off
set -u
export TMPDIR=/tmp
run() { VX_TIMELINE=1 VX_TIMELINE_PART=3 timeout 200 python3 profiles/timeline.py --format esvo --hot 0 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: part 3 us', d['us_in_service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'trips', d['loop_trips_per_wave'][2])"; }
run "sorted"
VX_SORTED_NOPART=1 run "no partition"
VX_SORTED_NOPART=1 VX_SORTED_NOREC=1 run "no partition, no records"
VX_SORTED_NOPART=1 VX_SORTED_NOREC=1 VX_HOLD_RESOLVED=2 run "no partition, no records, no look-ahead in mid-pass"
VX_SORTED=0 run "unsorted"

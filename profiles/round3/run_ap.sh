#!/bin/bash
# round 3, pass AP: SORTED builds with a block re-sorted every fourth frame: whole GPU suite, C3 by frames in flight, the refill part
set -u
export TMPDIR=/tmp
O=gpurun_out/r3ap; mkdir -p $O; rm -f $O/*
timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed|rc=|Error" $O/pytest.log | cut -c1-300
run() { VX_TIMELINE=1 VX_TIMELINE_PART=$2 timeout 200 python3 profiles/timeline.py --format esvo --hot 0 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: part $2 us', d['us_in_service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'trips', d['loop_trips_per_wave'][2], 'cycles/trip', d['cycles_per_trip_mean'])"; }
run "sorted, every 4th" 3 | tee $O/sorted4.txt
run "sorted, every 4th" 0 | tee -a $O/sorted4.txt
VX_SORT_EVERY_FRAME=1 run "sorted, every frame" 3 | tee -a $O/sorted4.txt
for fif in 2; do for so in 1 0; do for f in csvo esvo; do VX_SORTED=$so timeout 300 python3 bench.py --format $f --no-cpu-baseline --no-sd500 --repeats 9 --frames-in-flight $fif 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fif $fif sorted $so $f', d['value'], d['ms_per_step'], d['roofline']['kernel_exclusive_ms'], d['roofline']['kernel_exclusive_ms_timed_policy'])"; done; done; done | tee -a $O/sorted4.txt

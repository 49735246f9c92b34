#!/bin/bash
# round 3, pass AL: SORTED builds -- the queue hands out blocks of four sub-tiles, a wave renders a block in four passes whose pixels last frame's costs
# chose (VX_SORTED=0: the unsorted builds): whole GPU suite, C3 both ways, the timeline
set -u
export TMPDIR=/tmp
O=gpurun_out/r3al; mkdir -p $O; rm -f $O/*
timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed|rc=|Error" $O/pytest.log | cut -c1-300
for i in 1 2; do for so in 0 1; do for f in csvo esvo; do VX_SORTED=$so timeout 300 python3 bench.py --format $f --no-cpu-baseline --no-sd500 --repeats 9 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('sorted $so $f', d['value'], d['ms_per_step'], d['roofline']['kernel_exclusive_ms'], d['roofline']['kernel_exclusive_ms_timed_policy'])"; done; done; done | tee $O/sorted.txt
for so in 0 1; do for f in csvo esvo; do VX_SORTED=$so VX_TIMELINE=1 timeout 200 python3 profiles/timeline.py --format $f --hot 0 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('sorted $so $f timeline: trips', d['loop_trips_per_wave'][2], 'cycles/trip', d['cycles_per_trip_mean'], 'service us', d['us_in_service_phases_per_wave'][2], 'phases', d['service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'kernel', d['kernel_us'])"; done; done | tee -a $O/sorted.txt

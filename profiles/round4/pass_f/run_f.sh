#!/bin/bash
# round 4, pass F: walk phases that serve nobody but their walkers (the lanes of a sub-tile stay in lockstep)
set -u
export TMPDIR=/tmp
O=gpurun_out/r4f; mkdir -p $O; rm -f $O/*
export VX_FOREIGN_MIN=1
timeout 900 python -m pytest tests/test_baseline_c4_c5.py tests/test_hip_parity.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -3 $O/pytest.txt
timeout 600 python profiles/configs_bench.py --format csvo --configs C4-d13 C4 C5 > $O/configs_csvo.json 2> $O/configs_csvo.err
grep -h '"config"' $O/configs_csvo.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], d['ms_per_frame'], d['rays_led_into_a_voxel_per_frame'], d['of_which_started_over'], d['excursion_phases_per_frame'], d['iterations_on_bytes_per_frame'])
" | tee -a $O/summary.txt
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'us per wave p10/p50/p90:', d['us_in_service_phases_per_wave'][1:4], 'phases', d['service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'kernel', d['kernel_us'], 'cycles/trip', d['cycles_per_trip_mean'], 'trips', d['loop_trips_per_wave'][2], 'loop share', d['loop_share_of_wave_life'][2], 'tail', d['tail_us_per_wave'][2:5])"; }
for part in 0 5; do
    VX_TIMELINE_PART=$part VX_TIMELINE=1 timeout 300 python3 profiles/timeline.py --format csvo --depth 14 --width 3840 --height 2160 --hot 0 2>/dev/null | tail -n 1 | line "csvo d14 part $part" | tee -a $O/parts.txt
done

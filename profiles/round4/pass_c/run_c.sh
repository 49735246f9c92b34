#!/bin/bash
# round 4, pass C: the walk phases without their atomics (excursion counters only on request): deep-CSVO configurations, foreign_min 1 / 8 / 40
set -u
export TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O; rm -f $O/*
for fm in 1 8 40; do
  VX_FOREIGN_MIN=$fm timeout 600 python profiles/configs_bench.py --format csvo --configs C4-d13 C4 C5 > $O/configs_csvo_fm$fm.json 2> $O/configs_csvo_fm$fm.err
done
grep -h '"config"' $O/configs_*.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['format'], d['config'], d['ms_per_frame'], d['rays_led_into_a_voxel_per_frame'], d['of_which_started_over'], d['excursion_phases_per_frame'], d['iterations_on_bytes_per_frame'])
" | tee $O/summary.txt
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'us per wave p10/p50/p90:', d['us_in_service_phases_per_wave'][1:4], 'phases', d['service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'kernel', d['kernel_us'], 'cycles/trip', d['cycles_per_trip_mean'], 'trips', d['loop_trips_per_wave'][2], 'loop share', d['loop_share_of_wave_life'][2], 'tail', d['tail_us_per_wave'][2:5])"; }
for fm in 1 40; do for part in 0 5; do
    VX_FOREIGN_MIN=$fm VX_TIMELINE_PART=$part VX_TIMELINE=1 timeout 300 python3 profiles/timeline.py --format csvo --depth 14 --width 3840 --height 2160 --hot 0 2>/dev/null | tail -n 1 | line "csvo d14 fm$fm part $part" | tee -a $O/parts.txt
done; done

#!/bin/bash
# round 3, pass G: where a wave's time goes at 4K on the depth-13 terrain, CSVO (excursions into voxels) against ESVO
set -u
O=gpurun_out/r3g; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
for f in csvo esvo; do for part in 0 1 2 3 4 5; do
  VX_TIMELINE_PART=$part VX_TIMELINE=1 timeout 300 python3 profiles/timeline.py --format $f --depth 13 --width 3840 --height 2160 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f part', $part, 'us per wave p10/p50/p90:', d['us_in_service_phases_per_wave'][1:4], 'phases', d['service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'kernel', d['kernel_us'], 'cycles/trip', d['cycles_per_trip_mean'], 'trips', d['loop_trips_per_wave'][2], 'loop share', d['loop_share_of_wave_life'][2], 'tail', d['tail_us_per_wave'][2:5])" >> $O/parts_d13.txt
done; done
cat $O/parts_d13.txt
for fm in 8 16 32 48; do VX_FOREIGN_MIN=$fm timeout 300 python3 profiles/configs_bench.py --format csvo --configs C4-d13 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('foreign_min', $fm, d['ms_per_frame'], d['excursion_phases_per_frame'])"; done
for sm in 32 48; do VX_SERVICE_MIN=$sm timeout 300 python3 profiles/configs_bench.py --format csvo --configs C4-d13 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('service_min', $sm, d['ms_per_frame'], d['excursion_phases_per_frame'])"; done

#!/bin/bash
# round 3, pass C (the hand-scheduled loop is in): where a wave's service-phase time goes now (per part), and whether the thresholds
# that were tuned for the old loop (service_min 32, refill_min 4) are still the right ones.
set -u
O=gpurun_out/r3c; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
for f in csvo esvo; do for part in 0 1 2 3 4; do
  VX_TIMELINE_PART=$part VX_TIMELINE=1 timeout 200 python3 profiles/timeline.py --format $f 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f part', $part, 'us per wave p10/p50/p90:', d['us_in_service_phases_per_wave'][1:4], 'phases', d['service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'kernel', d['kernel_us'], 'cycles/trip', d['cycles_per_trip_mean'])" >> $O/parts.txt
done; done
cat $O/parts.txt
for f in csvo esvo; do
  timeout 600 python3 profiles/sweep.py --format $f --rounds 5 --steps 20 --configs "s=32,r=4" "s=24,r=4" "s=40,r=4" "s=48,r=4" "s=32,r=8" "s=40,r=8" "s=32,r=2" "s=36,r=4" "s=32,r=4,f=1" "s=40,r=4,f=1" 2>&1 | grep -v "^counters" > $O/sweep_$f.txt
  cat $O/sweep_$f.txt
done

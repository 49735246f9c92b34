#!/bin/bash
# round 3, pass D: leaf tests of opaque blocks without the sample (RenderParams::opaque_*): parity (whole GPU suite), then bench.py with the set
# and without it (VX_NO_OPAQUE_SET=1: every leaf test samples, as before), alternately; then the thresholds again.
set -u
O=gpurun_out/r3d; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 4 $O/pytest.log
for i in 1 2 3; do for f in csvo esvo; do
  VX_NO_OPAQUE_SET=1 timeout 300 python3 bench.py --format $f --no-cpu-baseline --repeats 11 2>/dev/null | tail -n 1 > $O/sampled_${f}_$i.json
  timeout 300 python3 bench.py --format $f --no-cpu-baseline --repeats 11 2>/dev/null | tail -n 1 > $O/opaque_${f}_$i.json
done; done
python3 - $O <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + '/*_*_[0-9].json')):
    try:
        d = json.loads(open(f).read()); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline'].get('kernel_exclusive_ms'))
    except Exception as e: print(f, 'ERR', e)
PY
for f in csvo esvo; do for part in 0 1 2; do
  VX_TIMELINE_PART=$part VX_TIMELINE=1 timeout 200 python3 profiles/timeline.py --format $f 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f part', $part, 'us per wave p10/p50/p90:', d['us_in_service_phases_per_wave'][1:4], 'phases', d['service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'kernel', d['kernel_us'], 'cycles/trip', d['cycles_per_trip_mean'])" >> $O/parts.txt
done; done
cat $O/parts.txt
for f in csvo; do
  timeout 600 python3 profiles/sweep.py --format $f --rounds 5 --steps 20 --configs "s=32,r=4" "s=40,r=4" "s=48,r=4" "s=52,r=4" "s=56,r=4" "s=60,r=4" "s=48,r=8" "s=56,r=8" 2>&1 | grep -v "^counters" > $O/sweep_$f.txt
  cat $O/sweep_$f.txt
done

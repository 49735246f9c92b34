#!/bin/bash
# round 4, pass T5: a shaded hit's normal-map and colour samples side by side (texture_lod_pair): suite, bench, the finished-rays part of a wave's life
set -u
export TMPDIR=/tmp
O=gpurun_out/r4t; mkdir -p $O
timeout 1500 python -u -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -n 4 | tee $O/pytest_t5.txt
for rep in 1 2 3; do for fmt in csvo esvo; do
  timeout 600 python bench.py --format $fmt --no-cpu-baseline --no-extras > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('$fmt: in flight', d['ms_per_step'], d['value'], 'one at a time (HIP bracket)', d['roofline'].get('kernel_exclusive_ms'))" | tee -a $O/pair.txt
done; done
for part in 0 2; do
VX_TIMELINE_PART=$part VX_TIMELINE=1 timeout 300 python3 profiles/timeline.py --format csvo --hot 1 2>/dev/null | tail -n 1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('csvo C3 part $part: us per wave p10/p50/p90', d['us_in_service_phases_per_wave'][1:4], 'lifetime', d['mean_wave_lifetime_us'], 'kernel', d['kernel_us'])" | tee -a $O/pair.txt
done

#!/bin/bash
# round 4, pass R20: the refill's ticket settled before the phase's pixel stores
set -u
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
timeout 900 python -u -m pytest tests/test_hip_parity.py tests/test_baseline_configs.py -m gpu -x -q --timeout 300 -k "moving or sizes or edges or cost_ordered or c3 or C3 or versions or sharded" 2>&1 | tail -2 | tee -a $O/settle.txt
for rep in 1 2; do for fmt in csvo esvo; do
  timeout 600 python bench.py --format $fmt --no-cpu-baseline --no-extras > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('$fmt: in flight', d['ms_per_step'], 'one at a time (HIP bracket)', d['roofline'].get('kernel_exclusive_ms'))" | tee -a $O/settle.txt
done; done
VX_TIMELINE_PART=3 VX_TIMELINE=1 timeout 300 python3 profiles/timeline.py --format csvo --hot 1 2>/dev/null | tail -n 1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('csvo C3 part 3 (refill): us per wave p10/p50/p90', d['us_in_service_phases_per_wave'][1:4], 'lifetime', d['mean_wave_lifetime_us'], 'kernel', d['kernel_us'])" | tee -a $O/settle.txt

#!/bin/bash
# "fuller lanes": a wave leaves its loop when VX_SERVICE_MIN lanes wait (64 = lockstep); the freed lanes get the next rays -- the same sub-tile's shadow
# rays or the next sub-tile's primary rays -- while the stragglers finish. At HEAD, the benchmark's moving camera, two frames in flight and one.
B="python bench.py --steps 20 --warmup 5 --repeats 9 --no-cpu-baseline --no-extras --sustained-seconds 1"
j() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$1', 'burst', d['burst']['ms_per_step'], 'sustained', d['sustained']['ms_per_step'], 'exclusive', d['roofline']['kernel_exclusive_ms'])"; }
for fmt in csvo esvo; do for m in 64 60 56 48 40 32; do VX_SERVICE_MIN=$m $B --format $fmt 2>/dev/null | j ${fmt}_service_min_$m; done; done

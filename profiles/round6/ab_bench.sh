#!/bin/bash
# a short A/B of the headline: bench.py without its secondary blocks, both formats; prints burst / sustained / exclusive ms per frame
B="python3 bench.py --steps 20 --warmup 5 --repeats 15 --no-cpu-baseline --no-extras --sustained-seconds 2"
j() { python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$1', 'burst', d['burst']['ms_per_step'], 'sustained', d['sustained']['ms_per_step'], 'exclusive', d['roofline']['kernel_exclusive_ms'], 'one-frame wall', d['config'].get('one_frame_at_a_time_wall_ms'))"; }
for rep in 1 2; do for fmt in csvo esvo; do $B --format $fmt 2>/dev/null | j ${fmt}; done; done

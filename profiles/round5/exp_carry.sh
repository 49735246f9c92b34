#!/bin/bash
# (the code this script drove was built in commit e67b562 and reverted: profiles/round5/pass_a/carry.txt)
# lockstep with one early exit per batch: VX_CARRY = stragglers carried into the next batch (0 = plain lockstep)
B="python bench.py --steps 20 --warmup 5 --repeats 9 --no-cpu-baseline --no-extras --sustained-seconds 1"
j() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$1', 'burst', d['burst']['ms_per_step'], 'sustained', d['sustained']['ms_per_step'], 'exclusive', d['roofline']['kernel_exclusive_ms'])"; }
for fmt in csvo esvo; do for k in 0 2 4 8 12 16 24 32; do VX_CARRY=$k $B --format $fmt 2>/dev/null | j ${fmt}_carry_$k; done; done

#!/bin/bash
# one frame at a time: issue priority for the waves on the dearest sub-tiles (VX_HOT_PRIO=n: the first 1/n of the cost-ordered table)
B="python bench.py --steps 20 --warmup 5 --repeats 9 --no-cpu-baseline --no-extras --sustained-seconds 0"
j() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$1', d['value'], d['ms_per_step'], 'exclusive', d['roofline']['kernel_exclusive_ms'], d['roofline']['kernel_exclusive_ms_timed_policy'])"; }
for fmt in csvo esvo; do
for n in 0 4 8 16 32 64; do VX_HOT_PRIO=$n $B --format $fmt 2>/dev/null | j ${fmt}_prio_$n; done
done

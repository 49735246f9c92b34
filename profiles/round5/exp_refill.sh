#!/bin/bash
# (measured at commit 488851f, "experiment: refill threshold in the product build": VX_EXP_REFILL_MIN existed there and nowhere since -- at HEAD nothing
# reads it and every setting prints the same numbers. To repeat the comparison today: VX_REFILL_MIN with VX_LIB_DIR=voxel-rs_amd/lib/lib_tl, the measurement build.)
# pure batches: idle lanes are refilled only when VX_EXP_REFILL_MIN of them are free (4 = round 4's; 64 = a sub-tile at a time: no batch mixes the next
# sub-tile's primary rays with this one's shadow rays)
B="python bench.py --steps 20 --warmup 5 --repeats 15 --no-cpu-baseline --no-extras --sustained-seconds 2"
j() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$1', 'burst', d['burst']['ms_per_step'], 'sustained', d['sustained']['ms_per_step'], 'exclusive', d['roofline']['kernel_exclusive_ms'])"; }
for rep in 1 2; do for fmt in csvo esvo; do for r in 4 32 64; do VX_EXP_REFILL_MIN=$r $B --format $fmt 2>/dev/null | j ${fmt}_refill_$r; done; done; done
for r in 4 64; do VX_EXP_REFILL_MIN=$r python profiles/configs_bench.py --format csvo --configs C2 C4-d13 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if 'config' in d: print('refill $r', d['config'], d['ms_per_frame'])"; done

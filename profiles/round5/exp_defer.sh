#!/bin/bash
# (drove code that was built, measured and reverted: commit 7f12221 -- check it out to run this)
# shadow rays deferred (PersistentArgs::shadow_queue): off / on, the number of records a wave collects before it traces them, the lanes that must
# have ended before it serves / refills. The measurement build reads the knobs.
export VX_LIB_DIR=$PWD/voxel-rs_amd/lib/lib_tl
B="python bench.py --steps 20 --warmup 5 --repeats 15 --no-cpu-baseline --no-extras --sustained-seconds 2"
j() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$1', 'burst', d['burst']['ms_per_step'], 'sustained', d['sustained']['ms_per_step'], 'exclusive', d['roofline']['kernel_exclusive_ms'])"; }
for rep in 1 2; do for fmt in esvo csvo; do
  VX_DEFER_SHADOWS=0 $B --format $fmt 2>/dev/null | j ${fmt}_lockstep
  for sw in 64 128 192 256; do for sv in 48 32 16; do
    VX_DEFER_SHADOWS=1 VX_DEFER_SWITCH=$sw VX_DEFER_SERVICE=$sv $B --format $fmt 2>/dev/null | j ${fmt}_defer_switch${sw}_service${sv}
  done; done
done; done

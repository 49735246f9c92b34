#!/bin/bash
# (drove code that was built, measured and reverted: commit 7f12221 -- check it out to run this)
# shadow rays deferred, larger sessions (a queue of 1088 records a wave), C3 and 4K depth 13
export VX_LIB_DIR=$PWD/voxel-rs_amd/lib/lib_tl
B="python bench.py --steps 20 --warmup 5 --repeats 15 --no-cpu-baseline --no-extras --sustained-seconds 2"
j() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$1', 'burst', d['burst']['ms_per_step'], 'sustained', d['sustained']['ms_per_step'], 'exclusive', d['roofline']['kernel_exclusive_ms'])"; }
for rep in 1 2; do
  VX_DEFER_SHADOWS=0 $B --format esvo 2>/dev/null | j esvo_lockstep
  for sw in 256 512 1024; do for sv in 40 32 24; do
    VX_DEFER_SHADOWS=1 VX_DEFER_SWITCH=$sw VX_DEFER_SERVICE=$sv $B --format esvo 2>/dev/null | j esvo_defer_switch${sw}_service${sv}
  done; done
done
for cfg in "0 256 32" "1 256 32" "1 512 32" "1 1024 32" "1 1024 24"; do set -- $cfg
  VX_DEFER_SHADOWS=$1 VX_DEFER_SWITCH=$2 VX_DEFER_SERVICE=$3 python profiles/configs_bench.py --format esvo --configs C4-d13 C5-d13 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l)
    if 'config' in d: print('defer $1 switch $2 service $3', d['config'], d['ms_per_frame'])"
done

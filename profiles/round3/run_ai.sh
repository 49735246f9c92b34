#!/bin/bash
# round 3, pass AI: image-only kernels that also fold away the order table and the screen sharding (a scratch build in voxel-rs_amd/lib_exp) against HEAD
set -u
export TMPDIR=/tmp
for i in 1 2 3; do for L in lib lib_exp; do for f in csvo esvo; do VX_LIB_DIR=voxel-rs_amd/$L timeout 300 python3 bench.py --format $f --no-cpu-baseline --no-sd500 --repeats 9 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L $f', d['value'], d['ms_per_step'], d['roofline']['kernel_exclusive_ms'], d['roofline']['kernel_exclusive_ms_timed_policy'])"; done; done; done

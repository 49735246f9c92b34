#!/bin/bash
# round 3, pass W: five waves per SIMD (12-level stack with a 16-bit third plane: 7.5 KB a wave, 96 registers) with the hand-scheduled loop
set -u
export TMPDIR=/tmp
O=gpurun_out/r3w; mkdir -p $O; rm -f $O/*
VX_FIVE_WAVES=1 timeout 600 python3 -m pytest tests -m gpu -x -q -k "kernel_versions or heightfield_frame or full_size" 2>&1 | tail -n 2 | cut -c1-200
for f in esvo csvo; do timeout 900 python3 profiles/sweep.py --format $f --rounds 5 --steps 20 --configs "W=0" "W=1" "W=1,f=3" "W=0,w=12" 2>&1 | grep -v "^counters" ; done | tee $O/five_waves.txt

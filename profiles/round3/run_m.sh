#!/bin/bash
# round 3, pass M: full lockstep (service_min 64) and the refill threshold
set -u
O=gpurun_out/r3m; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
for f in csvo esvo; do
timeout 900 python3 profiles/sweep.py --format $f --rounds 5 --steps 20 --configs "s=56,r=4" "s=62,r=4" "s=63,r=4" "s=64,r=4" "s=64,r=1" "s=64,r=16" "s=64,r=32" "s=64,r=48" "s=64,r=64" "s=64,r=64,f=3" "s=64,r=4,f=3" "s=64,r=4,f=1" 2>&1 | grep -v "^counters" | tee $O/sweep_$f.txt
done

#!/bin/bash
# round 3, pass N: service phases of one kind (VX_PURE_PHASES): parity, then the service threshold with and without
set -u
O=gpurun_out/r3n; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests -m gpu -x -q -k "kernel_versions or heightfield_frame or golden or deep_world or inside or full_size or translucent or cost_ordered or edges" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 3 $O/pytest.log | cut -c1-200
for f in csvo esvo; do
timeout 900 python3 profiles/sweep.py --format $f --rounds 5 --steps 20 --configs "P=0,s=64" "P=1,s=64" "P=1,s=60" "P=1,s=56" "P=1,s=52" "P=1,s=48" "P=1,s=40" "P=1,s=32" "P=1,s=56,r=16" "P=1,s=48,r=16" 2>&1 | grep -v "^counters" | tee $O/sweep_$f.txt
done

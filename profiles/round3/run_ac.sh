#!/bin/bash
# round 3, pass AC: the order table's classes now that a note is one atomic per sub-tile and phase: VX_COST_FLOOR (a sub-tile is noted from this
# many iterations of its longest ray on) x VX_COST_STEP (iterations per class, sixteen classes); the kernel alone, one frame at a time
set -u
export TMPDIR=/tmp
O=gpurun_out/r3ac; mkdir -p $O; rm -f $O/*
for f in esvo csvo; do for fl in 64 32 16 0; do for st in 16 8 24; do
  VX_COST_FLOOR=$fl VX_COST_STEP=$st timeout 300 python3 bench.py --format $f --no-cpu-baseline --no-sd500 --repeats 5 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f floor $fl step $st', d['value'], d['ms_per_step'], d['roofline']['kernel_exclusive_ms'], d['roofline']['kernel_exclusive_ms_timed_policy'])"
done; done; done | tee $O/cost_classes.txt

#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r3y; mkdir -p $O; rm -f $O/*
for fm in 16 24 28 32 40 48; do for c in C4-d13 C4; do VX_FOREIGN_MIN=$fm timeout 600 python3 profiles/configs_bench.py --format csvo --configs $c 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('hold 1 foreign_min', $fm, d['config'], d['ms_per_frame'], 'phases', d['excursion_phases_per_frame'])"; done; done | tee $O/foreign_min_hold.txt

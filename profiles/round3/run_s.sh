#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r3s; mkdir -p $O
for fm in 32 48 56 64; do for c in C4-d13 C4; do VX_FOREIGN_MIN=$fm timeout 600 python3 profiles/configs_bench.py --format csvo --configs $c 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('foreign_min', $fm, d['config'], d['ms_per_frame'], d['excursion_phases_per_frame'])"; done; done | tee $O/foreign_min.txt

#!/bin/bash
# round 3, pass U: a cap on the walk inside a voxel (VX_MAX_WALK: longer walks are given up and their pixels rendered on the bytes later)
set -u
export TMPDIR=/tmp
O=gpurun_out/r3u; mkdir -p $O; rm -f $O/*
for mw in 1000000 12 8 6 5 4 3; do for c in C4-d13 C4; do VX_MAX_WALK=$mw timeout 600 python3 profiles/configs_bench.py --format csvo --configs $c 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('max_walk', $mw, d['config'], d['ms_per_frame'], 'phases', d['excursion_phases_per_frame'], 'given up', d['of_which_started_over'], 'iterations on bytes', d['iterations_on_bytes_per_frame'])"; done; done | tee $O/max_walk.txt

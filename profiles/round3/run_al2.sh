#!/bin/bash
# round 3, pass AL2: SORTED builds: frames in flight x persistent waves per CU (a launch's share of the device), both formats
set -u
export TMPDIR=/tmp
O=gpurun_out/r3al2; mkdir -p $O; rm -f $O/*
for f in esvo csvo; do for fif in 3 4 5 6 8; do for w in 6 8 10 12 16; do VX_SORTED=1 VX_WAVES_PER_CU=$w timeout 300 python3 bench.py --format $f --no-cpu-baseline --no-sd500 --repeats 7 --frames-in-flight $fif 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('sorted $f fif $fif waves/cu $w', d['value'], d['ms_per_step'])"; done; done; done | tee $O/fif_waves.txt

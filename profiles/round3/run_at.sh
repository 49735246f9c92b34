#!/bin/bash
# round 3, pass AT: sorted passes for the builds with the excursion code too (VX_SORTED_DEEP=1, a build since removed): parity on the deep worlds, 4K depth 13 / C4 / C5, CSVO
set -u
export TMPDIR=/tmp
O=gpurun_out/r3at; mkdir -p $O; rm -f $O/*
VX_SORTED_DEEP=1 timeout 1200 python3 -m pytest tests -m gpu -x -q -k "deep_world or inside or c4 or c5 or streamed" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed|rc=|Error" $O/pytest.log | cut -c1-300
for sd in 0 1; do for c in C4-d13 C4 C5; do VX_SORTED_DEEP=$sd timeout 600 python3 profiles/configs_bench.py --format csvo --configs $c 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('sorted_deep $sd', d['config'], d['ms_per_frame'], 'phases', d['excursion_phases_per_frame'], 'given up', d['of_which_started_over'])"; done; done | tee $O/sorted_deep.txt

#!/bin/bash
# round 3, pass AM: SORTED builds with their defaults (four frames in flight, a launch on half the wave slots, blocks noted by their passes' sum): whole GPU
# suite, the bench lines both ways
set -u
export TMPDIR=/tmp
O=gpurun_out/r3am; mkdir -p $O; rm -f $O/*
timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed|rc=|Error" $O/pytest.log | cut -c1-300
for i in 1 2; do for so in 1 0; do for f in csvo esvo; do VX_SORTED=$so timeout 300 python3 bench.py --format $f --no-cpu-baseline --repeats 9 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('sorted $so $f', d['value'], d['ms_per_step'], d['roofline']['kernel_exclusive_ms'], d['roofline']['kernel_exclusive_ms_timed_policy'], 'sd500', d['shadow_distance_500']['value'])"; done; done; done | tee $O/sorted_defaults.txt
timeout 300 python3 bench.py --no-cpu-baseline --no-sd500 --repeats 9 --force-sharded 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('sharded', d['value'], d['ms_per_step'])" | tee -a $O/sorted_defaults.txt

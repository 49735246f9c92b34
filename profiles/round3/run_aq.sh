#!/bin/bash
# round 3, pass AQ: sorted passes under a camera that moves every frame (profiles/moving_camera.py): VX_SORTED=2 sorted always, 1 (default) only for a view
# that has not moved since the stream's last frame, 0 never; then the whole GPU suite and the bench lines
set -u
export TMPDIR=/tmp
O=gpurun_out/r3aq; mkdir -p $O; rm -f $O/*
for f in esvo csvo; do for deg in 0 0.1 2; do for so in 2 1 0; do VX_SORTED=$so timeout 300 python3 profiles/moving_camera.py --format $f --degrees $deg 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f sorted $so degrees/frame', d['degrees_per_frame'], 'ms/frame', d['ms_per_frame'])"; done; done; done | tee $O/moving.txt
timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed|rc=|Error" $O/pytest.log | cut -c1-300
for f in csvo esvo; do timeout 300 python3 bench.py --format $f --no-cpu-baseline --repeats 9 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench $f', d['value'], d['ms_per_step'], d['roofline']['kernel_exclusive_ms'], d['roofline']['kernel_exclusive_ms_timed_policy'], 'sd500', d['shadow_distance_500']['value'])"; done | tee -a $O/moving.txt

#!/bin/bash
# round 3, pass AU: sorted passes on other still views of the C3 scene (yaw / pitch of the view direction), both ways; and the cost-ordered queue's test
set -u
export TMPDIR=/tmp
O=gpurun_out/r3au; mkdir -p $O; rm -f $O/*
python3 -m pytest tests -m gpu -x -q -k 'cost_ordered or kernel_versions' 2>&1 | grep -E 'passed|failed' | cut -c1-200
for v in "0 -0.35" "90 -0.35" "200 -0.1" "300 -0.8" "45 0.1"; do set -- $v; for so in 1 0; do VX_SORTED=$so timeout 300 python3 profiles/moving_camera.py --format esvo --degrees 0 --start $1 --pitch $2 --frames 300 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('yaw $1 pitch $2 sorted $so ms/frame', d['ms_per_frame'])"; done; done | tee $O/views.txt

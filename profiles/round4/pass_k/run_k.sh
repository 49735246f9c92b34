#!/bin/bash
# round 4, pass K: passes sorted by the last frame's costs REPROJECTED into this frame's view (sort_passes_kernel): moving and still views
set -u
export TMPDIR=/tmp
O=gpurun_out/r4k; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_baseline_configs.py tests/test_sharding.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -5 $O/pytest.txt
for f in csvo esvo; do
  for srt in 1 0; do
    for deg in 0 0.1 0.25 1.0; do
      VX_SORTED=$srt timeout 300 python profiles/moving_camera.py --format $f --degrees $deg --frames 400 2>/dev/null | tail -1 | sed "s/^/sorted=$srt /" | tee -a $O/moving.txt
    done
  done
done
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d.get('moving_camera'))"

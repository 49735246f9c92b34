#!/bin/bash
# round 4, pass M: picker latency (zero-copy small batches), image build on up to 64 threads (commit_s), streaming at HEAD, one frame at a time (timeline)
set -u
export TMPDIR=/tmp
O=gpurun_out/r4m; mkdir -p $O; rm -f $O/*
timeout 600 python -m pytest tests/test_baseline_configs.py tests/test_hip_parity.py -m gpu -x -q -k "picker or c1 or raycast or physics or mapper" > $O/pytest_picker.txt 2>&1; echo "pytest rc $?" >> $O/pytest_picker.txt; tail -3 $O/pytest_picker.txt
timeout 300 python profiles/picker_bench.py > $O/picker.json 2>$O/picker.err; cat $O/picker.json
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], 'picker', d.get('picker'))" | tee -a $O/summary.txt
for f in csvo esvo; do
  timeout 900 python profiles/configs_bench.py --format $f --configs C4 > $O/configs_$f.json 2> $O/configs_$f.err; head -1 $O/configs_$f.json | tee -a $O/summary.txt
  timeout 900 python profiles/stream_bench.py --format $f --scene-depth 14 --radius 40 --width 3840 --height 2160 --frames 80 --capacity-mb 3000 > $O/stream_d14_$f.json 2> $O/stream_d14_$f.err; tail -1 $O/stream_d14_$f.json | tee -a $O/summary.txt
done
for hot in 1 0; do
  VX_TIMELINE=1 timeout 300 python3 profiles/timeline.py --format csvo --depth 12 --hot $hot 2>/dev/null | tail -n 1 | tee -a $O/timeline_c3.txt
done

#!/bin/bash
# round 2: CSVO worlds of at most 12 levels: rays that start inside a voxel listed and run on the world's bytes when the wave is done (kForeignRerun)
# against the excursion inside the render loop (VX_FOREIGN_RERUN=0): the whole GPU suite, then C3
set -u
O=gpurun_out/rerun; mkdir -p $O; rm -rf $O/*
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 600 python profiles/sweep.py --format csvo --depth 12 --configs "R=0,f=2" "R=1,f=2" "R=0,f=1" "R=1,f=1" --rounds 5 --steps 20 2>&1 | grep -v "^counters\|amdgpu.ids" > $O/sweep_c3.txt
for f in csvo esvo; do timeout 300 python bench.py --format $f --no-cpu-baseline --repeats 11 2>/dev/null | tail -n 1 > $O/bench_$f.json; done
grep -E "passed|failed" $O/pytest.log; cat $O/sweep_c3.txt
python3 - <<'PY'
import json
for f in ('csvo','esvo'):
    d=json.loads(open('gpurun_out/rerun/bench_%s.json'%f).read()); print(f, d['value'], d['ms_per_step'], d['roofline']['kernel_exclusive_ms'], d['roofline']['frac'])
PY

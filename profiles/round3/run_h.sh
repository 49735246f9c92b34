#!/bin/bash
# round 3, pass H: the hand-scheduled loop for wide images and 16-level stacks: the whole GPU suite, then the deep configurations
set -u
O=gpurun_out/r3h; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 4 $O/pytest.log | cut -c1-300
for f in csvo esvo; do timeout 1500 python3 profiles/configs_bench.py --format $f --configs C3 C4-d13 C4 C5 > $O/configs_$f.json 2> $O/configs_$f.err; python3 - $O/configs_$f.json <<'PY'
import sys, json
for l in open(sys.argv[1]):
    d = json.loads(l)
    if 'config' in d: print(d['format'], d['config'], d['ms_per_frame'], d['Mrays_per_s'], d['Giterations_per_s'])
PY
done

#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r3r; mkdir -p $O; rm -f $O/*
timeout 900 python3 -m pytest tests -m gpu -x -q -k "kernel_versions or deep_world or inside or c4 or c5 or streamed" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 3 $O/pytest.log | cut -c1-200
timeout 1500 python3 profiles/configs_bench.py --format csvo --configs C3 C4-d13 C4 C5 > $O/configs_csvo.json 2> $O/configs_csvo.err
python3 - $O/configs_csvo.json <<'PY'
import sys, json
for l in open(sys.argv[1]):
    d = json.loads(l)
    if 'config' in d: print(d['format'], d['config'], d['ms_per_frame'], d['Mrays_per_s'], d['Giterations_per_s'], d.get('excursion_phases_per_frame'), d.get('of_which_started_over'))
PY
for fm in 16 24 32 48; do VX_FOREIGN_MIN=$fm timeout 300 python3 profiles/configs_bench.py --format csvo --configs C4-d13 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('foreign_min', $fm, d['ms_per_frame'], d['excursion_phases_per_frame'])"; done

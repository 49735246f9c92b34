#!/bin/bash
# round 2, GPU pass S: entry of a possible PUSH requested at the top of the iteration (Trav::step_with, kAhead) against the build before
# (voxel-rs_amd/lib_ab/base = HEAD~): parity, then the bench with each library alternately
set -u
O=gpurun_out/r2s; mkdir -p $O; rm -rf $O/*
timeout 900 python -m pytest tests -m gpu -x -q -k "kernel_versions or heightfield_frame or golden or deep_world or inside" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
for i in 1 2 3; do for f in esvo csvo; do
  VX_LIB_DIR=voxel-rs_amd/lib_ab/base timeout 300 python bench.py --format $f --no-cpu-baseline --repeats 11 2>/dev/null | tail -n 1 > $O/base_${f}_$i.json
  timeout 300 python bench.py --format $f --no-cpu-baseline --repeats 11 2>/dev/null | tail -n 1 > $O/new_${f}_$i.json
done; done
tail -n 3 $O/pytest.log
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2s/*_*.json')):
    try:
        d=json.loads(open(f).read()); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline'].get('kernel_exclusive_ms'))
    except Exception as e: print(f,'ERR',e)
PY

#!/bin/bash
# A/B of two builds of the libraries on one box: voxel-rs_amd/lib_ab (the build before; VX_LIB_DIR) against voxel-rs_amd/lib (this one): a parity
# subset on this one, then bench.py with each library alternately (default mode: two frames in flight), then this one's timeline.
#   usage: profiles/round3/ab.sh [tag] [formats]
set -u
TAG=${1:-ab}
FMTS=${2:-"csvo esvo"}
O=gpurun_out/$TAG; mkdir -p $O; rm -rf $O/*
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests -m gpu -x -q -k "kernel_versions or heightfield_frame or golden or deep_world or inside or full_size" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -n 3 $O/pytest.log
for i in 1 2 3; do for f in $FMTS; do
  VX_LIB_DIR=voxel-rs_amd/lib_ab timeout 300 python3 bench.py --format $f --no-cpu-baseline --repeats 11 2>/dev/null | tail -n 1 > $O/base_${f}_$i.json
  timeout 300 python3 bench.py --format $f --no-cpu-baseline --repeats 11 2>/dev/null | tail -n 1 > $O/new_${f}_$i.json
done; done
for f in $FMTS; do VX_TIMELINE=1 timeout 200 python3 profiles/timeline.py --format $f 2>/dev/null | tail -n 1 > $O/timeline_$f.json; done
python3 - $O <<'PY'
import json, glob, sys
o = sys.argv[1]
for f in sorted(glob.glob(o + '/*_*_[0-9].json')):
    try:
        d = json.loads(open(f).read()); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline'].get('kernel_exclusive_ms'))
    except Exception as e: print(f, 'ERR', e)
for f in sorted(glob.glob(o + '/timeline_*.json')):
    try:
        d = json.loads(open(f).read()); print(f.split('/')[-1], 'kernel_us', d['kernel_us'], 'cycles/trip', d['cycles_per_trip_mean'], 'loop share', d['loop_share_of_wave_life'][2], 'trips p50', d['loop_trips_per_wave'][2], 'service us p50', d['us_in_service_phases_per_wave'][2], 'clock', d['clock_mhz'][2])
    except Exception as e: print(f, 'ERR', e)
PY

#!/bin/bash
# A/B of two builds of the libraries on one box: voxel-rs_amd/lib_ab/$1 (the build before) against voxel-rs_amd/lib (this one) -- parity of this
# one, then bench.py with each library alternately, then the service-phase timeline of this one.  usage: run_ab.sh <base dir name> [quick]
set -u
BASE=${1:-v1}
O=gpurun_out/ab; mkdir -p $O; rm -rf $O/*
timeout 900 python -m pytest tests -m gpu -x -q -k "kernel_versions or heightfield_frame or golden or deep_world or inside or sharded or baseline_c3" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
for i in 1 2 3; do for f in esvo csvo; do
  VX_LIB_DIR=voxel-rs_amd/lib_ab/$BASE timeout 300 python bench.py --format $f --no-cpu-baseline --repeats 11 2>/dev/null | tail -n 1 > $O/base_${f}_$i.json
  timeout 300 python bench.py --format $f --no-cpu-baseline --repeats 11 2>/dev/null | tail -n 1 > $O/new_${f}_$i.json
done; done
for i in 1 2; do
  VX_LIB_DIR=voxel-rs_amd/lib_ab/$BASE timeout 300 python bench.py --format csvo --force-sharded --no-cpu-baseline --repeats 11 2>/dev/null | grep '^{' | tail -n 1 > $O/base_sharded_$i.json
  timeout 300 python bench.py --format csvo --force-sharded --no-cpu-baseline --repeats 11 2>/dev/null | grep '^{' | tail -n 1 > $O/new_sharded_$i.json
done
for part in 0 1 2 3 4; do VX_TIMELINE_PART=$part VX_TIMELINE=1 timeout 200 python profiles/timeline.py --format esvo --hot 7 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('part', $part, 'us per wave p10/p50/p90:', d['us_in_service_phases_per_wave'][1:4], 'phases', d['service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'kernel', d['kernel_us'])" >> $O/parts_esvo.txt; done
tail -n 3 $O/pytest.log
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/*_*_*.json')):
    try:
        d=json.loads(open(f).read()); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline'].get('kernel_exclusive_ms'))
    except Exception as e: print(f,'ERR',e)
PY
cat $O/parts_esvo.txt

#!/bin/bash
# round 3, pass T: rank 0's assembly as a workgroup per tile; bench.py --force-sharded against the plain line
set -u
export TMPDIR=/tmp
O=gpurun_out/r3t; mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests -m gpu -x -q -k "sharded or tile or gather or c5 or rgba8 or cabi" 2>&1 | tail -n 3 | cut -c1-200
for i in 1 2; do for fmt in rgba8 rgba32f; do timeout 300 python3 bench.py --no-cpu-baseline --no-sd500 --force-sharded --gather-format $fmt --repeats 7 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('sharded $fmt', d['value'], d['ms_per_step'], d['config']['sharded_frame_identical_to_whole_render'])"; done; timeout 300 python3 bench.py --no-cpu-baseline --no-sd500 --repeats 7 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['value'], d['ms_per_step'])"; done | tee $O/sharded_vs_plain.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline --no-sd500 --force-sharded --repeats 5 > $O/sharded.json 2> $O/sharded.err
python3 - <<'PY' | tee gpurun_out/r3t/kernels.txt
import csv,glob
for f in glob.glob('gpurun_out/r3t/trace/*/*kernel_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:4]:
        print(r['Name'][:60].replace('(anonymous namespace)::',''), r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY

#!/bin/bash
# round 4, pass R: the sub-tile queue's dispensers hand out REGIONS of the screen (one per XCD) instead of every eighth sub-tile
set -u
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O; rm -f $O/*
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -3 $O/pytest.txt
for fmt in csvo esvo; do
  timeout 900 python bench.py --format $fmt --no-cpu-baseline > $O/bench_$fmt.json 2> $O/bench_$fmt.err
  python3 -c "
import json; d=json.loads(open('$O/bench_$fmt.json').read().strip().split('\n')[-1])
print('$fmt', 'moving', d['value'], d['ms_per_step'], 'kernel_exclusive', d['roofline'].get('kernel_exclusive_ms'), 'still', d.get('still_view',{}).get('ms_per_step'))" | tee -a $O/summary.txt
  timeout 900 python profiles/configs_bench.py --format $fmt --configs C2 C4-d13 C4 C5 > $O/configs_$fmt.json 2>/dev/null
  grep -h '"config"' $O/configs_$fmt.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$fmt', d['config'], d['ms_per_frame'])
" | tee -a $O/summary.txt
done

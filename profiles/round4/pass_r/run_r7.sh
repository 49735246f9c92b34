#!/bin/bash
# round 4, pass R7: the stretches by launch (frames in flight: an eighth of the launch; one at a time: a tile, or a sub-tile under the cost order): suite, bench, configs
set -u
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
timeout 1500 python -u -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -n 5 > $O/pytest.txt; tail -3 $O/pytest.txt
for fmt in csvo esvo; do
  timeout 900 python bench.py --format $fmt --no-cpu-baseline > $O/bench_$fmt.json 2> $O/bench_$fmt.err
  python3 -c "
import json; d=json.loads(open('$O/bench_$fmt.json').read().strip().split('\n')[-1])
print('$fmt', 'moving', d['value'], d['ms_per_step'], 'kernel_exclusive', d['roofline'].get('kernel_exclusive_ms'), 'still', d.get('still_view',{}).get('ms_per_step'))" | tee -a $O/summary7.txt
  timeout 900 python profiles/configs_bench.py --format $fmt --configs C2 C4-d13 C4 C5 > $O/configs_$fmt.json 2>/dev/null
  grep -h '"config"' $O/configs_$fmt.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$fmt', d['config'], d['ms_per_frame'])
" | tee -a $O/summary7.txt
done

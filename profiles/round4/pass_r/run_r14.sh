#!/bin/bash
# round 4, pass R14: the state that is kept (tiles numbered in strips of 8 columns, tile lists with the stride; stretches of a tile): suite, bench, configs
set -u
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
timeout 1500 python -u -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -n 5 > $O/pytest14.txt; tail -3 $O/pytest14.txt
for fmt in csvo esvo; do
  timeout 900 python bench.py --format $fmt > $O/bench14_$fmt.json 2> $O/bench14_$fmt.err
  python3 -c "
import json; d=json.loads(open('$O/bench14_$fmt.json').read().strip().split('\n')[-1])
print('$fmt', 'moving', d['value'], d['ms_per_step'], 'kernel_exclusive', d['roofline'].get('kernel_exclusive_ms'), 'still', d.get('still_view',{}).get('ms_per_step'), 'sd500', d.get('shadow_distance_500',{}).get('ms_per_step'))" | tee -a $O/summary14.txt
  timeout 900 python profiles/configs_bench.py --format $fmt > $O/configs14_$fmt.json 2>/dev/null
  grep -h '"config"' $O/configs14_$fmt.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$fmt', d['config'], d['ms_per_frame'])
" | tee -a $O/summary14.txt
done

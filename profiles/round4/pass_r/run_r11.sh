#!/bin/bash
# round 4, pass R11: tile numbering 2 (the stride) as the default -- suite, bench (plain and forced-sharded), configs; numbering 1 beside it on the configs
set -u
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
timeout 1500 python -u -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -n 5 > $O/pytest11.txt; tail -3 $O/pytest11.txt
VX_TILE_NUMBERING=1 timeout 900 python -u -m pytest tests/test_hip_parity.py tests/test_baseline_configs.py -m gpu -x -q --timeout 300 -k "moving or sizes or edges or cost_ordered or c3 or C3 or versions or sharded" 2>&1 | tail -2 | tee -a $O/summary11.txt
for fmt in csvo esvo; do
  timeout 900 python bench.py --format $fmt --no-cpu-baseline > $O/bench_$fmt.json 2> $O/bench_$fmt.err
  python3 -c "
import json; d=json.loads(open('$O/bench_$fmt.json').read().strip().split('\n')[-1])
print('$fmt', 'moving', d['value'], d['ms_per_step'], 'kernel_exclusive', d['roofline'].get('kernel_exclusive_ms'), 'still', d.get('still_view',{}).get('ms_per_step'))" | tee -a $O/summary11.txt
  for num in 2 1; do
  VX_TILE_NUMBERING=$num timeout 900 python profiles/configs_bench.py --format $fmt --configs C2 C4-d13 C4 C5 2>/dev/null | grep -h '"config"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$fmt numbering $num', d['config'], d['ms_per_frame'])
" | tee -a $O/summary11.txt
  done
done
timeout 900 python bench.py --format csvo --no-cpu-baseline --no-extras --force-sharded > $O/bench_sharded.json 2> $O/bench_sharded.err
python3 -c "
import json; d=json.loads(open('$O/bench_sharded.json').read().strip().split('\n')[-1])
print('csvo forced-sharded', d['value'], d['ms_per_step'])" | tee -a $O/summary11.txt

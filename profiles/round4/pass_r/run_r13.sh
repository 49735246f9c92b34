#!/bin/bash
# round 4, pass R13: a tile list (forced-sharded, one rank) numbered as it lies (0: Morton order, every rank's share) or with the stride (2); the plain run beside it
set -u
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
for rep in 1 2; do
for num in 0 2; do
VX_TILE_NUMBERING=$num timeout 900 python bench.py --format csvo --no-cpu-baseline --no-extras --force-sharded > $O/b.json 2>/dev/null
python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('csvo forced-sharded, list numbering $num', d['value'], d['ms_per_step'])" | tee -a $O/sharded.txt
done
timeout 900 python bench.py --format csvo --no-cpu-baseline --no-extras > $O/b.json 2>/dev/null
python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('csvo plain', d['value'], d['ms_per_step'], d['roofline'].get('kernel_exclusive_ms'))" | tee -a $O/sharded.txt
done

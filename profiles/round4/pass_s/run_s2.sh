#!/bin/bash
# round 4, pass S2: frames in flight 2 / 3 / 4 under the new tile order (C3, moving camera)
set -u
export TMPDIR=/tmp
O=gpurun_out/r4s; mkdir -p $O
for rep in 1 2; do for fif in 2 3 4; do
  timeout 600 python bench.py --format csvo --no-cpu-baseline --no-extras --frames-in-flight $fif > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('csvo frames in flight $fif:', d['ms_per_step'], d['value'])" | tee -a $O/fif.txt
done; done

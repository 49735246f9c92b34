#!/bin/bash
# C4 (a still 4K view of the depth-14 terrain, two frames in flight) under the product build's knobs that do not change a pixel: how the tiles are numbered,
# the lockstep threshold, frames in flight. (The defaults were chosen at C3.)
run() { python3 profiles/round6/deep_frames.py --format $1 --frames 24 --sweep "0:2" 2>&1 | grep "^{" | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$1', '$2', d['ms_per_frame'])"; }
for f in esvo csvo; do
  run $f default
  for tn in 0 1 2; do VX_TILE_NUMBERING=$tn run $f tile_numbering=$tn; done
  for sm in 1 8 16; do VX_SERVICE_MIN=$sm run $f service_min=$sm; done
  run $f default-again
done

#!/bin/bash
# round 4, pass S4: streaming (C4 streamed: depth-14 terrain, radius 40, 4K) and presenting every frame, at HEAD
set -u
export TMPDIR=/tmp
O=gpurun_out/r4s; mkdir -p $O
for f in csvo esvo; do
  timeout 900 python profiles/stream_bench.py --format $f --scene-depth 14 --radius 40 --width 3840 --height 2160 --frames 80 --capacity-mb 3000 > $O/stream_d14_$f.json 2> $O/stream_d14_$f.err; tail -1 $O/stream_d14_$f.json | cut -c1-1500
done
timeout 600 python profiles/present_bench.py > $O/present.json 2> $O/present.err; tail -3 $O/present.json | cut -c1-1200

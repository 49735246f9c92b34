#!/bin/bash
# The streamer's initial fill of the depth-14 terrain (radius 40), a few times per format. HSA_ENABLE_SDMA=0/1 in front of it tells what the DMA engines cost.
for i in 1 2 3 4; do
  for f in csvo esvo; do
    python3 profiles/stream_bench.py --format $f --scene-depth 14 --radius 40 --width 3840 --height 2160 --frames 2 2>/dev/null | grep -o '"initial_fill": {[^}]*}' | head -1 | sed "s/^/$f /"
  done
done

#!/bin/bash
# RenderParams::tile_numbering 1 (strips of eight columns, the default) against 2 (places a golden-ratio stride apart) at C3, C4 and C5's frame size.
B="python3 bench.py --steps 20 --warmup 5 --repeats 15 --no-cpu-baseline --no-extras --sustained-seconds 2"
j() { python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$1', 'burst', d['burst']['ms_per_step'], 'sustained', d['sustained']['ms_per_step'], 'exclusive', d['roofline']['kernel_exclusive_ms'])"; }
for rep in 1 2; do for tn in 1 2; do for fmt in csvo esvo; do VX_TILE_NUMBERING=$tn $B --format $fmt 2>/dev/null | j "C3 numbering=$tn ${fmt}"; done; done; done
run() { python3 profiles/round6/deep_frames.py --format $1 --size $3 --frames 16 --sweep "0:1 0:2" 2>&1 | grep "^{" | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$3', '$1', '$2', 'in flight', d['frames_in_flight'], d['ms_per_frame'])"; }
for rep in 1 2; do for f in esvo csvo; do for tn in 1 2; do VX_TILE_NUMBERING=$tn run $f numbering=$tn 3840x2160; done; done; done
for f in esvo csvo; do for tn in 1 2; do VX_TILE_NUMBERING=$tn run $f numbering=$tn 7680x4320; done; done

#!/bin/bash
# Numbering 1's strip width (the measurement build's VX_TILE_STRIP; the product's is 8 tiles) at C4 and C5's frame size, CSVO.
run() { VX_LIB_DIR=voxel-rs_amd/lib/lib_tl python3 profiles/round6/deep_frames.py --format $1 --size $3 --frames 16 --sweep "0:2" 2>&1 | grep "^{" | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$3', '$1', '$2', d['ms_per_frame'])"; }
for rep in 1 2; do for w in 8 4 16 32; do VX_TILE_NUMBERING=1 VX_TILE_STRIP=$w run csvo strip=$w 3840x2160; done; done
for w in 8 16 32; do VX_TILE_NUMBERING=1 VX_TILE_STRIP=$w run csvo strip=$w 7680x4320; done

#!/bin/bash
# round 6, step 1b: the waves-per-CU x frames-in-flight sweep at C4 (the measurement build: VX_WAVES_PER_CU is its knob), and the wave timeline of a C4 frame
# (one frame at a time): what a trip of the loop costs a wave at depth 14, how much of a wave's life its service phases and walks are.
set -u
out=gpurun_out/r6_c4_sweep
mkdir -p "$out"
export TMPDIR=/tmp
for fmt in csvo esvo; do
  VX_LIB_DIR=voxel-rs_amd/lib/lib_tl timeout -k 10 600 python3 profiles/round6/deep_frames.py --format $fmt --frames 16 \
     --sweep "16:1 16:2 16:3 16:4 12:1 12:2 12:4 8:1 8:2 8:4" > "$out/sweep_$fmt.txt" 2>&1
  grep '^{' "$out/sweep_$fmt.txt" | cut -c1-200
  VX_TIMELINE=1 timeout -k 10 300 python3 profiles/timeline.py --format $fmt --depth 14 --width 3840 --height 2160 > "$out/timeline_d14_$fmt.json" 2> "$out/timeline_d14_$fmt.err"
  VX_TIMELINE=1 timeout -k 10 300 python3 profiles/timeline.py --format $fmt --depth 12 --width 1920 --height 1080 > "$out/timeline_c3_$fmt.json" 2> "$out/timeline_c3_$fmt.err"
  tail -c 3000 "$out/timeline_d14_$fmt.json"
done

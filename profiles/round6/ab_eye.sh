#!/bin/bash
# (the experiment of commit ebcb2fe, reverted by faed70b: run it on a checkout of ebcb2fe; results: r6_ab_eye.txt)
# Primary rays on the eye's path (PersistentArgs::eye_*) against VX_EYE_PATH=0, same build: C3 (bench.py's headline without its secondary blocks) and C4
# (deep_frames.py: a still 4K view of the depth-14 terrain, one / two frames in flight), both formats.
B="python3 bench.py --steps 20 --warmup 5 --repeats 15 --no-cpu-baseline --no-extras --sustained-seconds 2"
j() { python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$1', 'burst', d['burst']['ms_per_step'], 'sustained', d['sustained']['ms_per_step'], 'exclusive', d['roofline']['kernel_exclusive_ms'], 'one-frame wall', d['config'].get('one_frame_at_a_time_wall_ms'))"; }
for rep in 1 2; do for eye in 1 0; do for fmt in csvo esvo; do VX_EYE_PATH=$eye $B --format $fmt 2>/dev/null | j "eye=$eye ${fmt}"; done; done; done
for eye in 1 0; do for fmt in csvo esvo; do
  VX_EYE_PATH=$eye python3 profiles/round6/deep_frames.py --format $fmt --frames 16 --sweep "0:1 0:2" 2>&1 | grep "^{" | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('eye=$eye $fmt C4 frames in flight', d['frames_in_flight'], 'ms', d['ms_per_frame'])"
done; done

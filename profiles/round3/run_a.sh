#!/bin/bash
# round 3, GPU pass A (baseline at the round's start + the evidence VERDICT r2 items 1 and 2 ask for):
#   1. the benched frame as a parity test (test_full_size_frames: asset textures, two frames in flight, shadow distance inf / 500)
#   2. bench.py, default mode, both formats (this box's baseline)
#   3. the same command under rocprofv3 --kernel-trace --stats (profile_fif2.sh)
#   4. valu_issue: cycles per wave64 VALU instruction at 1-4 waves per SIMD
#   5. timeline with shader-clock stamps: the clock render_persistent runs at and what a trip of its loop costs
set -u
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
python3 -m pytest tests/test_baseline_configs.py -x -q -m gpu -k full_size > gpurun_out/r3a/parity.log 2>&1; tail -3 gpurun_out/r3a/parity.log
for f in csvo esvo; do python3 bench.py --format $f > gpurun_out/r3a/bench_$f.json 2> gpurun_out/r3a/bench_$f.err; tail -c 1500 gpurun_out/r3a/bench_$f.json; done
profiles/round3/profile_fif2.sh csvo
profiles/round3/profile_fif2.sh esvo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue profiles/tools/valu_issue.hip && /tmp/valu_issue > gpurun_out/r3a/valu_issue.jsonl 2>&1; cat gpurun_out/r3a/valu_issue.jsonl
for f in csvo esvo; do VX_TIMELINE=1 python3 profiles/timeline.py --format $f > gpurun_out/r3a/timeline_$f.json 2>&1; tail -1 gpurun_out/r3a/timeline_$f.json; done

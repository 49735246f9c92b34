#!/bin/bash
# round 4: the rocprofv3 evidence for bench.py at HEAD -- one frame at a time (kernel trace + PMC passes, per format) and the timed mode (two frames in flight)
set -u
export TMPDIR=/tmp
for f in csvo esvo; do
  bash profiles/round4/profile_r4.sh $f > gpurun_out/prof_r4_$f.log 2>&1
  bash profiles/round4/profile_fif2.sh $f > gpurun_out/prof_r4_${f}_fif2.log 2>&1
done
tail -5 gpurun_out/prof_r4_csvo.log gpurun_out/prof_r4_csvo_fif2.log
# the first commit of the depth-14 terrain (worker pool)
for f in csvo; do timeout 900 python profiles/configs_bench.py --format $f --configs C4 2>/dev/null | head -1; done

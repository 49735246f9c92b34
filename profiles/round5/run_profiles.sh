#!/bin/bash
# round 5: the rocprofv3 evidence for bench.py at HEAD -- one frame at a time (kernel trace + PMC passes, per format) and the timed mode (two frames in
# flight) --, then profiles/round5/traffic.json (with the hash of the library sources it was measured on) and the copies that are committed
set -u
export TMPDIR=/tmp
for f in csvo esvo; do
  bash profiles/round5/profile_r5.sh $f > gpurun_out/prof_r5_$f.log 2>&1
  bash profiles/round5/profile_fif2.sh $f > gpurun_out/prof_r5_${f}_fif2.log 2>&1
done
tail -5 gpurun_out/prof_r5_csvo.log gpurun_out/prof_r5_csvo_fif2.log
python3 profiles/round5/make_traffic.py "${1:-?}" > gpurun_out/prof_r5_traffic.log 2>&1
mkdir -p gpurun_out/round5_copy
cp profiles/round5/traffic.json gpurun_out/round5_copy/traffic.json
for f in csvo esvo; do
  cp gpurun_out/prof_r5_$f/kernel_stats.csv gpurun_out/round5_copy/${f}_fif1_kernel_stats.csv
  cp gpurun_out/prof_r5_$f/summary.txt gpurun_out/round5_copy/${f}_fif1_rocprof_summary.txt
  cp gpurun_out/prof_r5_${f}_fif2/summary.txt gpurun_out/round5_copy/${f}_fif2_rocprof_summary.txt
  cp gpurun_out/prof_r5_${f}_fif2/bench_line.json gpurun_out/round5_copy/${f}_fif2_bench_line.json
  cp gpurun_out/prof_r5_${f}_fif2/kernel_stats.csv gpurun_out/round5_copy/${f}_fif2_kernel_stats.csv 2>/dev/null
  cp gpurun_out/prof_r5_${f}_fif2/block.json gpurun_out/round5_copy/${f}_fif2_block.json 2>/dev/null
done
ls -la gpurun_out/round5_copy

#!/bin/bash
# round 6: the rocprofv3 evidence for bench.py at HEAD -- C3 one frame at a time (kernel trace + PMC passes, per format), C3 in the timed mode (two frames in
# flight), C4 (3840x2160 on the depth-14 terrain) one frame at a time with the cache / VMEM counters (c4_counters.sh) --, then profiles/round6/traffic.json (keys
# csvo / esvo: C3; C4: {csvo, esvo}; with the hash of the library sources it was measured on) and the copies that are committed.
#   usage: profiles/round6/run_profiles.sh <commit>
set -u
export TMPDIR=/tmp
for f in csvo esvo; do
  bash profiles/round6/profile_r6.sh $f > gpurun_out/prof_r6_$f.log 2>&1
  bash profiles/round6/profile_fif2.sh $f > gpurun_out/prof_r6_${f}_fif2.log 2>&1
  SKIP_C3=1 bash profiles/round6/c4_counters.sh $f pmc > gpurun_out/r6_c4_$f.log 2>&1
done
tail -5 gpurun_out/prof_r6_csvo.log gpurun_out/prof_r6_csvo_fif2.log
python3 profiles/round6/make_traffic.py "${1:-?}" > gpurun_out/prof_r6_traffic.log 2>&1
tail -3 gpurun_out/prof_r6_traffic.log
mkdir -p gpurun_out/round6_copy
cp profiles/round6/traffic.json gpurun_out/round6_copy/traffic.json
for f in csvo esvo; do
  cp gpurun_out/prof_r6_$f/kernel_stats.csv gpurun_out/round6_copy/${f}_fif1_kernel_stats.csv
  cp gpurun_out/prof_r6_$f/summary.txt gpurun_out/round6_copy/${f}_fif1_rocprof_summary.txt
  cp gpurun_out/prof_r6_${f}_fif2/summary.txt gpurun_out/round6_copy/${f}_fif2_rocprof_summary.txt
  cp gpurun_out/prof_r6_${f}_fif2/bench_line.json gpurun_out/round6_copy/${f}_fif2_bench_line.json
  cp gpurun_out/prof_r6_${f}_fif2/kernel_stats.csv gpurun_out/round6_copy/${f}_fif2_kernel_stats.csv 2>/dev/null
  cp gpurun_out/prof_r6_${f}_fif2/block.json gpurun_out/round6_copy/${f}_fif2_block.json 2>/dev/null
  cp gpurun_out/r6_c4_$f/summary.txt gpurun_out/round6_copy/c4_${f}_rocprof_summary.txt
  find gpurun_out/r6_c4_$f/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/round6_copy/c4_${f}_kernel_stats.csv \;
done
ls -la gpurun_out/round6_copy

#!/bin/bash
# (round 6) the two passes of c4_counters.sh that its first run lost (FETCH_SIZE and WRITE_SIZE cannot be collected together): on their own, into the same directories
set -u
export TMPDIR=/tmp VX_FRAMES_IN_FLIGHT=1
for fmt in csvo esvo; do
  out=gpurun_out/r6_c4_$fmt
  mkdir -p "$out"
  run="python3 profiles/round6/deep_frames.py --format $fmt --frames 6 --warmup 2"
  timeout -k 10 420 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc12" -- $run > "$out/pmc12.log" 2>&1
  timeout -k 10 420 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc13" -- $run > "$out/pmc13.log" 2>&1
  python3 profiles/round6/pmc_summary.py "$out" > "$out/summary_write.txt"
  cp "$out/pmc.json" "$out/pmc_write.json"
  grep -A3 "pmc1[23]" "$out/summary_write.txt"
done

#!/bin/bash
# round 6: rocprofv3 of the command the driver runs -- `python3 bench.py --format <fmt>` with the default two frames
# in flight -- so that ms_per_step is reproducible from profiles/: per-kernel stats, and the start / end timestamps of consecutive
# render_persistent launches (two streams: launches overlap, one frame completes every (end[i+2] - end[i]) / 2).
#   usage: profiles/round6/profile_fif2.sh <csvo|esvo> [tag]
set -u
fmt=$1
tag=${2:-fif2}
out=gpurun_out/prof_r6_${fmt}_${tag}
mkdir -p "$out"
export TMPDIR=/tmp
args="--format $fmt --no-cpu-baseline --no-extras --steps 50 --warmup 10 --repeats 5 --sustained-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py $args > "$out/trace.log" 2>&1
python3 profiles/round6/fif2_blocks.py "$out" "$fmt" > "$out/summary.txt"
cat "$out/summary.txt" | cut -c1-400 | tail -45
grep -h '"metric"' "$out/trace.log" | tail -1 > "$out/bench_line.json"
python3 -c "import sys,json; d=json.loads(open('$out/bench_line.json').read()); print('bench under the tracer:', d['value'], d['ms_per_step'], d['roofline']['kernel_exclusive_ms'])"

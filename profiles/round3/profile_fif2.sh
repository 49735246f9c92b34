#!/bin/bash
# round 3: rocprofv3 of the command the driver runs -- `python3 bench.py --format <fmt>` with the default two frames
# in flight -- so that ms_per_step is reproducible from profiles/: per-kernel stats, and the start / end timestamps of consecutive
# render_persistent launches (two streams: launches overlap, one frame completes every (end[i+2] - end[i]) / 2).
#   usage: profiles/round3/profile_fif2.sh <csvo|esvo> [tag]
set -u
fmt=$1
tag=${2:-fif2}
out=gpurun_out/prof_r3_${fmt}_${tag}
mkdir -p "$out"
export TMPDIR=/tmp
args="--format $fmt --no-cpu-baseline --steps 50 --warmup 10 --repeats 5"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py $args > "$out/trace.log" 2>&1
python3 - "$out" "$fmt" <<'PY' > "$out/summary.txt"
import csv, glob, sys, json
out, fmt = sys.argv[1], sys.argv[2]
for f in glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True):
    print('== kernel stats (rocprofv3 --kernel-trace --stats -- python3 bench.py --format %s: the default TWO frames in flight)' % fmt)
    print(open(f).read())
    import shutil; shutil.copy(f, out + '/kernel_stats.csv')
rows = []
for f in glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'render_persistent' in r['Kernel_Name'] and '<2, false, true' not in r['Kernel_Name']:
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Stream_Id', r.get('Queue_Id', '?'))))
rows.sort()
# the timed blocks: launches that start less than 1 ms after their predecessor started
print('== consecutive render_persistent launches of one timed block (ns relative to the first; two streams, overlapping)')
blocks, cur = [], []
for r in rows:
    if cur and r[0] - cur[-1][0] > 2_000_000:
        blocks.append(cur); cur = []
    cur.append(r)
blocks.append(cur)
blocks = [b for b in blocks if len(b) == 50]
b = blocks[len(blocks) // 2] if blocks else max(blocks + [rows], key=len)
t0 = b[0][0]
print('launch,start_ns,end_ns,duration_ns,queue')
for i, (s, e, q) in enumerate(b[:24]):
    print('%d,%d,%d,%d,%s' % (i, s - t0, e - t0, e - s, q))
dur = [e - s for s, e, _ in b]
span = b[-1][1] - b[0][0]
per_frame = (b[-1][1] - b[1][1]) / (len(b) - 2) if len(b) > 2 else float('nan')
res = {'format': fmt, 'launches_in_block': len(b), 'blocks_found': len(blocks), 'mean_kernel_duration_ns': sum(dur) / len(dur), 'block_span_ns': span,
       'ns_per_frame_first_start_to_last_end': span / len(b), 'ns_per_frame_steady_state_end_to_end': per_frame,
       'overlap_factor': sum(dur) / span}
print('== block:', json.dumps(res))
json.dump(res, open(out + '/block.json', 'w'))
PY
cat "$out/summary.txt" | cut -c1-400 | tail -45
grep -h '"metric"' "$out/trace.log" | tail -1 > "$out/bench_line.json"
python3 -c "import sys,json; d=json.loads(open('$out/bench_line.json').read()); print('bench under the tracer:', d['value'], d['ms_per_step'], d['roofline']['kernel_exclusive_ms'])"

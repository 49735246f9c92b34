#!/usr/bin/env python3
"""Summary of a rocprofv3 --kernel-trace --stats run of bench.py in its timed mode (profile_fif2.sh): per-kernel stats and the start / end
timestamps of the consecutive render_persistent launches of one timed block.   usage: fif2_blocks.py <output dir> <csvo|esvo>"""
import csv, glob, sys, json
out, fmt = sys.argv[1], sys.argv[2]
for f in glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True):
    print('== kernel stats (rocprofv3 --kernel-trace --stats -- python3 bench.py --format %s: the default TWO frames in flight)' % fmt)
    print(open(f).read())
    import shutil; shutil.copy(f, out + '/kernel_stats.csv')
rows = []
for f in glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'render_persistent<3' in r['Kernel_Name'] or 'render_persistent<4' in r['Kernel_Name']:  # (the image kernels: the instrumented one walks the bytes)
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Stream_Id', r.get('Queue_Id', '?'))))
rows.sort()
# the timed blocks: launches that start less than 1 ms after their predecessor started
print('== consecutive render_persistent launches of one timed block (ns relative to the first; two streams, overlapping)')
gaps = sorted(b[0] - a[0] for a, b in zip(rows, rows[1:]))
typical = gaps[len(gaps) // 2] if gaps else 1
blocks, cur = [], []
for r in rows:
    if cur and r[0] - cur[-1][0] > 1.6 * typical and r[0] > cur[-1][1]:  # (a launch that starts after its predecessor ENDED, late: a barrier between blocks)
        blocks.append(cur); cur = []
    cur.append(r)
blocks.append(cur)
timed = [b for b in blocks if len(b) == 50]
b = timed[len(timed) // 2] if timed else max(blocks, key=len)
blocks = timed
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

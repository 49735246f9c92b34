#!/bin/bash
# forced-sharded, one rank: tile-list buffers (frames in flight) against frame streams
B="python bench.py --steps 20 --warmup 5 --repeats 15 --no-cpu-baseline --no-extras --sustained-seconds 0"
j() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$1', d['value'], d['ms_per_step'], d['config'].get('host_issue_ms_per_step'), d['roofline'].get('frames_in_flight'))"; }
$B 2>/dev/null | j plain_fif2
$B --force-sharded 2>/dev/null | j sharded_default
$B --force-sharded --frames-in-flight 3 --streams 2 2>/dev/null | j sharded_buf3_streams2
$B --force-sharded --frames-in-flight 4 --streams 2 2>/dev/null | j sharded_buf4_streams2
$B --force-sharded --frames-in-flight 6 --streams 2 2>/dev/null | j sharded_buf6_streams2
$B --force-sharded --frames-in-flight 4 --streams 3 2>/dev/null | j sharded_buf4_streams3

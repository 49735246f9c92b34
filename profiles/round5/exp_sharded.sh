#!/bin/bash
# where the forced-sharded path's 10 % go
B="python bench.py --steps 20 --warmup 5 --repeats 15 --no-cpu-baseline --no-extras --sustained-seconds 0"
j() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$1', d['value'], d['ms_per_step'], d['config'].get('host_issue_ms_per_step'), d['roofline'].get('frames_in_flight'))"; }
$B 2>/dev/null | j plain_fif2
$B --frames-in-flight 3 2>/dev/null | j plain_fif3
$B --force-sharded 2>/dev/null | j sharded_default
$B --force-sharded --frames-in-flight 2 2>/dev/null | j sharded_fif2
$B --force-sharded --frames-in-flight 4 2>/dev/null | j sharded_fif4
$B --force-sharded --gather-format rgba32f 2>/dev/null | j sharded_rgba32f
$B --force-sharded --gather torch 2>/dev/null | j sharded_torch
$B --force-sharded --gather-group 2 --frames-in-flight 4 2>/dev/null | j sharded_group2
$B --force-sharded --separate-calls 2>/dev/null | j sharded_separate_calls

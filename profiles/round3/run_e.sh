#!/bin/bash
# round 3, pass E: the N > 1 path of bench.py with one rank (--force-sharded): the library's exchange with its first-frames check, torch's,
# the fall-back from one to the other (simulated failure), both pixel formats; and the plain one-GPU line with its new fields.
set -u
O=gpurun_out/r3e; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
run() { name=$1; shift; timeout 600 python3 bench.py --no-cpu-baseline --repeats 7 "$@" > $O/$name.json 2> $O/$name.err; echo "== $name rc=$?"; tail -n 1 $O/$name.json | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print(d['value'], d['ms_per_step'], d['roofline']['kernel_exclusive_ms'], {k: c.get(k) for k in ('gather', 'gather_requested', 'gather_note', 'gather_format', 'sharded_frame_identical_to_whole_render', 'exchange_ms_per_gather_rank0')}, d.get('shadow_distance_500'))
print(c.get('per_rank'))"; grep -h "bench rank" $O/$name.err | head -3; }
run plain
run sharded_auto --force-sharded
run sharded_torch --force-sharded --gather torch
run sharded_fallback --force-sharded --simulate-gather-failure
run sharded_auto_f32 --force-sharded --gather-format rgba32f
run sharded_torch_f32 --force-sharded --gather torch --gather-format rgba32f
run sharded_auto_group2 --force-sharded --gather-group 2
timeout 600 python3 -m pytest tests -m gpu -x -q -k "gather or sharded or tile or present or cabi" 2>&1 | tail -3

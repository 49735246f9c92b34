#!/bin/bash
# round 3, pass AG: what the timeline instrumentation costs the kernel when it is off: voxel-rs_amd/lib_nt = a build without it (a.timeline folded to null:
# 123 -> 50 spilled SGPRs, 13 % fewer instructions in the ESVO image kernel)
set -u
export TMPDIR=/tmp
O=gpurun_out/r3ag; mkdir -p $O; rm -f $O/*
for i in 1 2; do for L in lib lib_nt; do for f in csvo esvo; do VX_LIB_DIR=voxel-rs_amd/$L timeout 300 python3 bench.py --format $f --no-cpu-baseline --no-sd500 --repeats 9 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L $f', d['value'], d['ms_per_step'], d['roofline']['kernel_exclusive_ms'], d['roofline']['kernel_exclusive_ms_timed_policy'])"; done; done; done | tee $O/code_size.txt
for L in lib lib_nt; do for c in C4-d13 C4; do VX_LIB_DIR=voxel-rs_amd/$L timeout 600 python3 profiles/configs_bench.py --format csvo --configs $c 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['config'], d['ms_per_frame'])"; done; done | tee -a $O/code_size.txt

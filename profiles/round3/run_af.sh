#!/bin/bash
# round 3, pass AF: does the kernel's code size matter? voxel-rs_amd/lib_p2 = a build whose sampler assumes power-of-two texture heights (REPEAT as a mask: the
# signed-modulo path, inlined at every sample, is a seventh of the ESVO kernel's code and a quarter of the deep-CSVO kernel's)
set -u
export TMPDIR=/tmp
O=gpurun_out/r3af; mkdir -p $O; rm -f $O/*
for i in 1 2; do for L in lib lib_p2; do for f in csvo esvo; do VX_LIB_DIR=voxel-rs_amd/$L timeout 300 python3 bench.py --format $f --no-cpu-baseline --no-sd500 --repeats 9 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L $f', d['value'], d['ms_per_step'], d['roofline']['kernel_exclusive_ms'], d['roofline']['kernel_exclusive_ms_timed_policy'])"; done; done; done | tee $O/code_size.txt
for L in lib lib_p2; do for c in C4-d13 C4; do VX_LIB_DIR=voxel-rs_amd/$L timeout 600 python3 profiles/configs_bench.py --format csvo --configs $c 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['config'], d['ms_per_frame'])"; done; done | tee -a $O/code_size.txt

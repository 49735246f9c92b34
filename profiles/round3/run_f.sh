#!/bin/bash
# round 3, pass F: the other configurations (C2, C3, 4K depth 13, C4, C5) with the round's kernels, both formats
set -u
O=gpurun_out/r3f; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp
for f in csvo esvo; do timeout 1500 python3 profiles/configs_bench.py --format $f --configs C2 C3 C4-d13 C4 C5 > $O/configs_$f.json 2> $O/configs_$f.err; cat $O/configs_$f.json | cut -c1-400; done

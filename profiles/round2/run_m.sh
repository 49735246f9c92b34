#!/bin/bash
# round 2, GPU pass M: five waves per SIMD (12-level stack, 16-bit third plane, 96 VGPRs) -- parity, then A/B
set -u
O=gpurun_out/r2m; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests -m gpu -x -q -k "kernel_versions or heightfield_frame or baseline" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
for f in esvo csvo; do
timeout 900 python profiles/sweep.py --format $f --depth 12 --configs "W=0,f=2" "W=1,f=2" "W=0,f=1" "W=1,f=1" "W=0,f=4" "W=1,f=4" --rounds 5 --steps 20 > $O/sweep_w5_$f.txt 2>&1
done
tail -n 3 $O/pytest.log; tail -n 8 $O/sweep_w5_esvo.txt; tail -n 8 $O/sweep_w5_csvo.txt

#!/bin/bash
# round 2, GPU pass N: batched service phases (BATCH kernels) -- parity, then A/B
set -u
O=gpurun_out/r2n; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests -m gpu -x -q -k "kernel_versions or heightfield_frame" > $O/pytest_first.log 2>&1; echo "rc=$?" >> $O/pytest_first.log
for f in esvo csvo; do
timeout 900 python profiles/sweep.py --format $f --depth 12 --configs "B=0,f=2" "B=1,f=2" "B=0,f=1" "B=1,f=1" "B=1,f=2,s=16" "B=1,f=2,s=40" "B=1,f=2,r=1" "B=1,f=2,r=16" --rounds 5 --steps 20 > $O/sweep_batch_$f.txt 2>&1
done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -n 3 $O/pytest_first.log; grep -E "passed|failed" $O/pytest_all.log; tail -n 8 $O/sweep_batch_esvo.txt; tail -n 8 $O/sweep_batch_csvo.txt

#!/bin/bash
# round 2, GPU pass G: expensive sub-tiles first -- parity, then A/B one frame at a time and two in flight
set -u
mkdir -p gpurun_out/r2g
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2g/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2g/pytest.log
for f in csvo esvo; do
timeout 600 python profiles/sweep.py --format $f --depth 12 --configs "h=0,f=1" "h=1,f=1" "h=0,f=2" "h=1,f=2" --rounds 4 --steps 20 > gpurun_out/r2g/sweep_hot_$f.txt 2>&1
done
timeout 600 python profiles/sweep.py --format esvo --depth 12 --tiles 8 --configs "h=0,f=8" "h=1,f=8" "h=0,f=3" "h=1,f=3" --rounds 4 --steps 40 > gpurun_out/r2g/sweep_hot_tiles8.txt 2>&1
tail -n 4 gpurun_out/r2g/pytest.log
for f in gpurun_out/r2g/sweep_*.txt; do echo "== $f"; tail -n 4 $f; done

#!/bin/bash
# round 2, GPU pass F: 16-level stack on the GPU (parity + A/B at depth 14), cost of carrying the excursion code (C3 CSVO), rocprofv3 evidence
set -u
mkdir -p gpurun_out/r2f
timeout 1500 python -m pytest tests -m gpu -x -q -k "deep_world or c4 or c5 or kernel_versions or heightfield" > gpurun_out/r2f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2f/pytest.log
timeout 600 python profiles/sweep.py --format esvo --depth 14 --width 3840 --height 2160 --configs "d=0" "d=1" --rounds 3 --steps 8 > gpurun_out/r2f/sweep_d14_esvo.txt 2>&1
timeout 600 python profiles/sweep.py --format csvo --depth 12 --configs "x=0" "x=1" --rounds 4 --steps 20 > gpurun_out/r2f/sweep_c3_csvo_noexc.txt 2>&1
timeout 600 python profiles/sweep.py --format esvo --depth 12 --configs "f=2" "f=2,s=24" --rounds 4 --steps 20 > gpurun_out/r2f/sweep_c3_esvo.txt 2>&1
for f in csvo esvo; do timeout 900 bash profiles/round2/profile_r2.sh $f > gpurun_out/r2f/profile_$f.log 2>&1; done
tail -4 gpurun_out/r2f/pytest.log
tail -3 gpurun_out/r2f/sweep_*.txt
tail -12 gpurun_out/r2f/profile_csvo.log

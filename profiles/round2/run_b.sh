#!/bin/bash
# round 2, GPU pass B: parity suite with the full-size C4 / C5 tests, C3 bench in both formats, configurations with excursion counts
set -u
mkdir -p gpurun_out/r2b
free -g > gpurun_out/r2b/host.txt; nproc >> gpurun_out/r2b/host.txt
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 ) > gpurun_out/r2b/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
for f in csvo esvo; do
  timeout 400 python bench.py --format $f > gpurun_out/r2b/bench_$f.json 2> gpurun_out/r2b/bench_$f.err
done
for f in csvo esvo; do
  timeout 900 python profiles/configs_bench.py --format $f --configs C2 C3 C4-d13 C4 C4-primary C5 > gpurun_out/r2b/configs_$f.json 2> gpurun_out/r2b/configs_$f.err
done
cat gpurun_out/r2b/host.txt
tail -25 gpurun_out/r2b/pytest.log
cat gpurun_out/r2b/bench_*.json gpurun_out/r2b/configs_*.json
tail -n 3 gpurun_out/r2b/*.err

#!/bin/bash
# round 2, GPU pass A: parity suite, C3 bench in both formats, the other configurations (incl. the static depth-14 terrain)
set -u
mkdir -p gpurun_out/r2a
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r2a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
for f in csvo esvo; do
  timeout 300 python bench.py --format $f --cpu-seconds 4 > gpurun_out/r2a/bench_$f.json 2> gpurun_out/r2a/bench_$f.err
done
for f in csvo esvo; do
  timeout 900 python profiles/configs_bench.py --format $f --configs C2 C3 C4-d13 C4 C4-primary C5 > gpurun_out/r2a/configs_$f.json 2> gpurun_out/r2a/configs_$f.err
done
tail -5 gpurun_out/r2a/pytest.log
cat gpurun_out/r2a/bench_*.json gpurun_out/r2a/configs_*.json
tail -3 gpurun_out/r2a/*.err

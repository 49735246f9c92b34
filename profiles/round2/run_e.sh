#!/bin/bash
# round 2, GPU pass E: streaming measurement (C4-shaped), forced-sharded bench output hygiene
set -u
mkdir -p gpurun_out/r2e
for f in csvo esvo; do
  timeout 600 python profiles/stream_bench.py --format $f --scene-depth 14 --radius 40 --frames 120 > gpurun_out/r2e/stream_d14_$f.json 2> gpurun_out/r2e/stream_d14_$f.err
done
timeout 400 python bench.py --format csvo --force-sharded --no-cpu-baseline > gpurun_out/r2e/forced.out 2> gpurun_out/r2e/forced.err
timeout 600 python -m pytest tests -m gpu -x -q -k "streaming or cabi or c_client" > gpurun_out/r2e/pytest.log 2>&1
cat gpurun_out/r2e/stream_d14_*.json
echo "--- forced-sharded stdout lines:"; wc -l gpurun_out/r2e/forced.out; tail -c 300 gpurun_out/r2e/forced.out
tail -5 gpurun_out/r2e/pytest.log
tail -n 3 gpurun_out/r2e/*.err

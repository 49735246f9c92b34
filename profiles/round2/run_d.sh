#!/bin/bash
# round 2, GPU pass D: the whole parity suite, C3 bench (plain and forced through the sharded path with the library's RCCL gather),
# presentation paths
set -u
mkdir -p gpurun_out/r2d
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 ) > gpurun_out/r2d/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2d/pytest.log
timeout 400 python bench.py --format csvo --cpu-seconds 3 > gpurun_out/r2d/bench_csvo.json 2> gpurun_out/r2d/bench_csvo.err
timeout 400 python bench.py --format csvo --force-sharded --no-cpu-baseline > gpurun_out/r2d/bench_csvo_forced_sharded.json 2> gpurun_out/r2d/bench_csvo_forced.err
timeout 400 python bench.py --format csvo --force-sharded --gather torch --no-cpu-baseline > gpurun_out/r2d/bench_csvo_forced_sharded_torch.json 2> gpurun_out/r2d/bench_csvo_forced_torch.err
for f in csvo esvo; do timeout 300 python profiles/present_bench.py --format $f > gpurun_out/r2d/present_$f.json 2> gpurun_out/r2d/present_$f.err; done
tail -22 gpurun_out/r2d/pytest.log
for f in gpurun_out/r2d/bench_*.json; do python3 -c "
import json,sys
d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['config'].get('gather'), d['config'].get('sharded_frame_identical_to_whole_render'), d['roofline']['kernel_exclusive_ms'], d['roofline']['frac'])"; done
cat gpurun_out/r2d/present_*.json
tail -n 4 gpurun_out/r2d/*.err

#!/bin/bash
# round 4, pass L: sorted passes removed; bench.py's headline is the moving camera's frame
set -u
export TMPDIR=/tmp
O=gpurun_out/r4l; mkdir -p $O; rm -f $O/*
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -5 $O/pytest.txt
for f in csvo esvo; do
  timeout 600 python bench.py --format $f > $O/bench_$f.json 2> $O/bench_$f.err
  python -c "
import json; d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); print('$f', 'value', d['value'], 'ms', d['ms_per_step'], 'still', d.get('still_view'), 'sd500', d.get('shadow_distance_500'), 'picker', d.get('picker'), 'excl', d['roofline']['kernel_exclusive_ms'], d['roofline']['kernel_exclusive_ms_timed_policy'], 'frac', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'] if d['cpu_baseline'] else None)" | tee -a $O/summary.txt
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --force-sharded --no-cpu-baseline > $O/bench_forced_sharded.json 2> $O/bench_forced_sharded.err
python -c "
import json; d=json.loads(open('$O/bench_forced_sharded.json').read().strip().splitlines()[-1]); print('forced sharded', d['value'], d['ms_per_step'], d['config'].get('gather'), d['config'].get('sharded_frame_identical_to_whole_render'))" | tee -a $O/summary.txt

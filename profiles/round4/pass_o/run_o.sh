#!/bin/bash
# round 4, pass O: one-wave assembly workgroups + one free wave slot for tile-list renders (forced-sharded against plain); the first commit of the
# depth-14 terrain with the image built in batches, huge-page hints and a pinned-bounce upload
set -u
export TMPDIR=/tmp
O=gpurun_out/r4o; mkdir -p $O; rm -f $O/*
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -4 $O/pytest.txt
for rep in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_plain_$rep.json 2> $O/bench_plain_$rep.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$rep bench.py --gpus 1 --force-sharded --no-cpu-baseline > $O/bench_forced_$rep.json 2> $O/bench_forced_$rep.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2952$rep bench.py --gpus 1 --force-sharded --no-cpu-baseline --gather-format rgba32f > $O/bench_forced_f32_$rep.json 2> $O/bench_forced_f32_$rep.err
done
python - <<'PY' | tee $O/summary.txt
import json, glob
for f in sorted(glob.glob('gpurun_out/r4o/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d['config'].get('gather'), d['config'].get('sharded_frame_identical_to_whole_render'), d['roofline']['frames_in_flight'])
    except Exception as e:
        print(f, 'failed', e)
PY
g++ -O2 -std=c++17 -pthread -Ivoxel-rs_amd/csrc/hip -o /tmp/image_build_time profiles/tools/image_build_time.cpp
python - <<'PY'
import sys
sys.path.insert(0, '.')
from _pkg import load_package
vra = load_package()
w = vra.World(2); w.build_heightfield(14)
w.frame(pad_words=0).tofile('/tmp/world14_2.bin')
PY
/tmp/image_build_time /tmp/world14_2.bin 2 2 16 16 | tee -a $O/image_build.txt
for f in csvo esvo; do
  timeout 900 python profiles/configs_bench.py --format $f --configs C4 2>/dev/null | head -1 | tee -a $O/summary.txt
done

#!/bin/bash
# round 4, pass N: the whole-world image build in batches, by the number of worker threads (host only), and C4's first commit
set -u
export TMPDIR=/tmp
O=gpurun_out/r4n; mkdir -p $O; rm -f $O/*
g++ -O2 -std=c++17 -pthread -Ivoxel-rs_amd/csrc/hip -o /tmp/image_build_time profiles/tools/image_build_time.cpp
python - <<'PY'
import sys
sys.path.insert(0, '.')
from _pkg import load_package
vra = load_package()
for fmt in (2, 1):
    w = vra.World(fmt); w.build_heightfield(14)
    w.frame(pad_words=0).tofile(f'/tmp/world14_{fmt}.bin')
PY
for fmt in 2 1; do /tmp/image_build_time /tmp/world14_$fmt.bin $fmt 2 16 32 64 16 | tee -a $O/image_build.txt; done
for f in csvo esvo; do
  timeout 900 python profiles/configs_bench.py --format $f --configs C4 > $O/configs_$f.json 2> $O/configs_$f.err; cat $O/configs_$f.json | tee -a $O/summary.txt
done

#!/bin/bash
# round 4, pass P: frames in flight 2 / 3 / 4 for the headline; the chunk walk of the whole-world image on 16 / 24 / 32 threads (batched build)
set -u
export TMPDIR=/tmp
O=gpurun_out/r4p; mkdir -p $O; rm -f $O/*
for fif in 2 3 4; do for f in csvo esvo; do
  timeout 600 python bench.py --format $f --no-cpu-baseline --no-extras --frames-in-flight $fif 2>/dev/null | tail -1 | python -c "
import sys, json; d=json.loads(sys.stdin.read()); print('$f frames in flight $fif:', d['value'], d['ms_per_step'], d['roofline']['kernel_exclusive_ms'])" | tee -a $O/frames_in_flight.txt
done; done
g++ -O2 -std=c++17 -pthread -Ivoxel-rs_amd/csrc/hip -o /tmp/image_build_time profiles/tools/image_build_time.cpp
python - <<'PY'
import sys
sys.path.insert(0, '.')
from _pkg import load_package
vra = load_package()
w = vra.World(2); w.build_heightfield(14)
w.frame(pad_words=0).tofile('/tmp/world14_2.bin')
PY
/tmp/image_build_time /tmp/world14_2.bin 2 2 16 24 32 48 16 | tee -a $O/image_build_threads.txt

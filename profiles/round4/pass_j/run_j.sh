#!/bin/bash
# round 4, pass J: after the split (kernels_render.hip / kernels_aux.hip / runtime.cpp / comm.cpp) and the pruning of the variants: the whole GPU suite,
# every configuration in both formats, the bench line
set -u
export TMPDIR=/tmp
O=gpurun_out/r4j; mkdir -p $O; rm -f $O/*
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -5 $O/pytest.txt
for f in csvo esvo; do
  timeout 900 python profiles/configs_bench.py --format $f --configs C2 C3 C4-d13 C4 C5 > $O/configs_$f.json 2> $O/configs_$f.err
done
grep -h '"config"' $O/configs_*.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['format'], d['config'], d['ms_per_frame'], d['Mrays_per_s'], d['rays_led_into_a_voxel_per_frame'], d['of_which_started_over'], d['excursion_phases_per_frame'])
" | tee $O/summary.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json

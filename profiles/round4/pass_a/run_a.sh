#!/bin/bash
# round 4, pass A: the lean walk inside voxels (walk_voxel_on_bytes) -- parity suite, then deep-CSVO configurations with a sweep of foreign_min
set -x
OUT=gpurun_out/r4a; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
tail -5 $OUT/pytest.txt
for fm in 1 16 40; do
  VX_FOREIGN_MIN=$fm timeout 600 python profiles/configs_bench.py --format csvo --configs C4-d13 C4 C5 > $OUT/configs_csvo_fm$fm.json 2> $OUT/configs_csvo_fm$fm.err
done
timeout 600 python profiles/configs_bench.py --format esvo --configs C4-d13 C4 C5 > $OUT/configs_esvo.json 2> $OUT/configs_esvo.err
grep -h '"config"' $OUT/configs_*.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['format'], d['config'], d['ms_per_frame'], d['rays_led_into_a_voxel_per_frame'], d['of_which_started_over'], d['excursion_phases_per_frame'], d['iterations_on_bytes_per_frame'])
"

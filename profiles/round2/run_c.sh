#!/bin/bash
# round 2, GPU pass C: how many lanes should go on the excursion together (CSVO images)
set -u
mkdir -p gpurun_out/r2c; rm -f gpurun_out/r2c/sweep.txt
for fm in 1 16 32; do
  VX_FOREIGN_MIN=$fm timeout 300 python profiles/configs_bench.py --format csvo --configs C3 C4-d13 --steps 20 2> gpurun_out/r2c/err_$fm.txt | grep config | sed "s/^/{\"foreign_min\": $fm} /" >> gpurun_out/r2c/sweep.txt
done
cat gpurun_out/r2c/sweep.txt | python3 -c "
import sys,json
for l in sys.stdin:
    a,b=l.split('} {',1); fm=json.loads(a+'}')['foreign_min']; r=json.loads('{'+b)
    print(fm, r['config'], r['ms_per_frame'], r['Mrays_per_s'], r['rays_led_into_a_voxel_per_frame'], r['of_which_started_over'], r['excursion_phases_per_frame'], r['iterations_on_bytes_per_frame'], r['cycles_per_excursion_phase'])
"
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "inside_voxels or heightfield or kernel_versions" 2>&1 | tail -3

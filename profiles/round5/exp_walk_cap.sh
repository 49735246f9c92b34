#!/bin/bash
# (the code this script drove was built in commit 00107c1 and reverted: profiles/round5/pass_a/walk_cap.txt)
# deep CSVO worlds: walks inside voxels in instalments of VX_EXP_WALK_CAP iterations per service phase (0 = whole, as round 4)
for cap in 0 2 3 4 6 8; do
  VX_EXP_WALK_CAP=$cap python profiles/configs_bench.py --format csvo --configs C4-d13 C4 C5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if 'config' in d: print('cap $cap', d['config'], d['ms_per_frame'], 'walks', d['rays_led_into_a_voxel_per_frame'], 'given up', d['of_which_started_over'], 'phases', d['excursion_phases_per_frame'], 'iters', d['iterations_on_bytes_per_frame'])"
done

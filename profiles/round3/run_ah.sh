#!/bin/bash
# round 3, pass AH: the timeline instrumentation as a kernel build of its own (template parameter TL; VX_TIMELINE=1 selects it): whole GPU suite, C3, the timeline tool, deep CSVO
set -u
export TMPDIR=/tmp
timeout 1800 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -E 'passed|failed|Error' | cut -c1-300
for f in csvo esvo; do python3 bench.py --format $f --no-cpu-baseline --no-sd500 --repeats 9 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'], d['roofline']['kernel_exclusive_ms'], d['roofline']['kernel_exclusive_ms_timed_policy'])"; done
for f in csvo esvo; do VX_TIMELINE=1 python3 profiles/timeline.py --format $f --hot 0 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f timeline: trips', d['loop_trips_per_wave'][2], 'cycles/trip', d['cycles_per_trip_mean'], 'service us', d['us_in_service_phases_per_wave'][2], 'phases', d['service_phases_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'], 'kernel', d['kernel_us'])"; done
VX_TIMELINE=1 python3 profiles/timeline.py --format csvo --depth 13 --width 3840 --height 2160 --hot 0 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('d13 timeline: trips', d['loop_trips_per_wave'][2], 'lifetime', d['mean_wave_lifetime_us'])"
for c in C4-d13 C4; do python3 profiles/configs_bench.py --format csvo --configs $c 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], d['ms_per_frame'])"; done

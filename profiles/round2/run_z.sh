#!/bin/bash
# round 2, GPU pass Z: everything once more at HEAD (after the load-ahead and ticket-ahead changes) -- parity suite, bench (both formats, forced-sharded), configurations, streaming
set -u
mkdir -p gpurun_out/r2z
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=6 ) > gpurun_out/r2z/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2z/pytest.log
for f in csvo esvo; do timeout 400 python bench.py --format $f > gpurun_out/r2z/bench_$f.json 2> gpurun_out/r2z/bench_$f.err; done
timeout 400 python bench.py --format csvo --force-sharded --no-cpu-baseline > gpurun_out/r2z/bench_csvo_forced_sharded.json 2> gpurun_out/r2z/forced.err
timeout 400 python bench.py --format csvo --frames-in-flight 4 --no-cpu-baseline > gpurun_out/r2z/bench_csvo_4_in_flight.json 2> gpurun_out/r2z/fif4.err
for f in csvo esvo; do timeout 900 python profiles/configs_bench.py --format $f --configs C2 C3 C4-d13 C4 C4-primary C5 > gpurun_out/r2z/configs_$f.json 2> gpurun_out/r2z/configs_$f.err; done
for f in csvo esvo; do for pl in 0 1; do timeout 600 python profiles/stream_bench.py --format $f --scene-depth 14 --radius 40 --frames 120 --pipelined $pl > gpurun_out/r2z/stream_d14_${f}_p$pl.json 2> gpurun_out/r2z/stream_${f}_p$pl.err; done; done
for f in csvo esvo; do timeout 300 python profiles/present_bench.py --format $f > gpurun_out/r2z/present_$f.json 2> gpurun_out/r2z/present_$f.err; done
timeout 300 python profiles/picker_bench.py > gpurun_out/r2z/picker.json 2> gpurun_out/r2z/picker.err
tail -n 12 gpurun_out/r2z/pytest.log
for f in gpurun_out/r2z/bench_*.json; do python3 -c "
import json
l=[x for x in open('$f').read().split('\n') if x.startswith('{')]
d=json.loads(l[-1]); r=d['roofline']; print('$f'.split('/')[-1], d['value'], d['ms_per_step'], 'excl', r['kernel_exclusive_ms'], 'frac', r['frac'], 'sust', r['sustained_GBps'], (d.get('cpu_baseline') or {}).get('value'), ((d.get('cpu_baseline') or {}).get('single_thread') or {}).get('value'))"; done
python3 -c "
import json
for f in ('csvo','esvo'):
    for l in open('gpurun_out/r2z/configs_%s.json'%f):
        d=json.loads(l)
        if 'config' in d: print(f, d['config'], d['ms_per_frame'], d['Mrays_per_s'], d['Giterations_per_s'], d['rays_led_into_a_voxel_per_frame'], d['of_which_started_over'])
"
cat gpurun_out/r2z/stream_d14_*.json | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['workload'][-5:], d.get('commit_mode'), 'host ms/step', d['host_ms_per_step_median'], 'with render call', d.get('host_ms_per_step_with_render_call_median'), 'apply', d['apply_ms_median'], 'commit', d['commit_ms_median'], 'fill s', d['initial_fill']['seconds'], 'settled Mrays/s', d['Mrays_per_s_settled'])"
tail -n 2 gpurun_out/r2z/picker.json

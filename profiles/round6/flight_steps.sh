for i in 1 2 3; do for f in csvo esvo; do
python3 profiles/stream_bench.py --format $f --scene-depth 14 --radius 40 --width 3840 --height 2160 --frames 80 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('$f', d['host_ms_per_step_median'], d['host_ms_per_step_max'], d['host_ms_dearest_steps'])
"
done; done

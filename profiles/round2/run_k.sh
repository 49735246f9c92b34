#!/bin/bash
# round 2, GPU pass K: pipelined commits (worker thread) -- parity, then the streaming bench in both modes; bench after the loop change
set -u
O=gpurun_out/r2k; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests -m gpu -x -q -k "pipelined or incremental or kernel_versions" > $O/pytest_new.log 2>&1; echo "rc=$?" >> $O/pytest_new.log
for f in esvo csvo; do for p in 0 1; do
  timeout 600 python profiles/stream_bench.py --format $f --pipelined $p > $O/stream_${f}_p$p.json 2> $O/stream_${f}_p$p.err
done; done
timeout 600 python bench.py --format csvo --no-cpu-baseline > $O/bench_csvo.json 2> $O/bench_csvo.err
timeout 600 python bench.py --format esvo --no-cpu-baseline > $O/bench_esvo.json 2> $O/bench_esvo.err
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -n 3 $O/pytest_new.log; tail -n 3 $O/pytest_all.log
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2k/stream_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], {k:d[k] for k in ('commit_mode','host_ms_per_step_median','host_ms_per_step_with_render_call_median','apply_ms_median','commit_ms_median','kernel_ms_streaming_median','kernel_ms_settled_median')})
    except Exception as e: print(f, 'ERR', e)
for f in sorted(glob.glob('gpurun_out/r2k/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline'].get('kernel_exclusive_ms'), d['roofline']['frac'])
    except Exception as e: print(f,'ERR',e)
PY

#!/bin/bash
# round 4, pass S3: the state at HEAD as the driver runs it -- smoke, the GPU suite, bench.py with its defaults (both formats)
set -u
export TMPDIR=/tmp
O=gpurun_out/r4s; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1500 python -u -m pytest tests -m gpu -x -q 2>&1 | tail -n 4 | tee $O/pytest.txt
timeout 900 python bench.py > $O/bench_csvo.json 2> $O/bench_csvo.err; tail -c 600 $O/bench_csvo.json
timeout 900 python bench.py --format esvo > $O/bench_esvo.json 2> $O/bench_esvo.err
python3 -c "
import json
for f in ('csvo','esvo'):
    d=json.loads(open('$O/bench_%s.json' % f).read().strip().split('\n')[-1])
    print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_exclusive_ms'], d['roofline']['issue']['frac_of_bound_timed_mode'], d['cpu_baseline']['value'], d['picker']['reference_batch_80']['us_per_call'])
" | tee $O/summary.txt

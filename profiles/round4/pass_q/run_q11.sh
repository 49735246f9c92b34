#!/bin/bash
# round 4, pass Q11: the whole GPU suite, the deep configurations and the benchmark at the state that is kept
set -u
export TMPDIR=/tmp
O=gpurun_out/r4q; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_full.txt 2>&1; echo "pytest rc $?" >> $O/pytest_full.txt; tail -3 $O/pytest_full.txt
for fmt in csvo esvo; do
timeout 900 python profiles/configs_bench.py --format $fmt --configs C4-d13 C4 C5 > $O/configs_$fmt.json 2>/dev/null
grep -h '"config"' $O/configs_$fmt.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$fmt', d['config'], d['ms_per_frame'])
" | tee -a $O/summary_kept.txt
done
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json

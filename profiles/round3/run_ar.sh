#!/bin/bash
# round 3, pass AR: sorted passes only for frames that cast shadow rays: C2 / C3 (synthetic textures), parity
set -u
export TMPDIR=/tmp
for f in csvo esvo; do timeout 600 python3 profiles/configs_bench.py --format $f --configs C2 C3 2>/dev/null | grep '"config"' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config'],'$f',d['ms_per_frame'],d['Mrays_per_s'])"; done
python3 -m pytest tests -m gpu -x -q -k 'kernel_versions or full_size' 2>&1 | grep -E 'passed|failed' | cut -c1-200

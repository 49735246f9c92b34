#!/bin/bash
# round 2, GPU pass T: load-ahead build (wide layout included): whole GPU suite, then every configuration with the build before and this one
set -u
O=gpurun_out/r2t; mkdir -p $O; rm -rf $O/*
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
for f in esvo csvo; do
  VX_LIB_DIR=voxel-rs_amd/lib_ab/base timeout 900 python profiles/configs_bench.py --format $f --configs C2 C3 C4-d13 C4 C5 2>/dev/null | grep '"config"' > $O/configs_base_$f.json
  timeout 900 python profiles/configs_bench.py --format $f --configs C2 C3 C4-d13 C4 C5 2>/dev/null | grep '"config"' > $O/configs_new_$f.json
done
grep -E "passed|failed" $O/pytest_all.log
python3 - <<'PY'
import json
for f in ('esvo','csvo'):
    for b in ('base','new'):
        try:
            rows=[json.loads(l) for l in open(f'gpurun_out/r2t/configs_{b}_{f}.json')]
            print(f, b, ' '.join(f"{r['config']}:{r['ms_per_frame']:.3f}ms/{r['Mrays_per_s']:.0f}" for r in rows))
        except Exception as e: print(f,b,'ERR',e)
PY
